"""openmpl_amd -- MI355X-native multi-view pose-lifting forward pass (OpenMPL hot path)."""
__version__ = "0.1.0"
