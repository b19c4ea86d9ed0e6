"""openmpl_amd -- MI355X-native multi-view pose-lifting forward pass (OpenMPL hot path)."""
__version__ = "0.1.0"


def check_device(device: int = -1, synchronize: bool = True):
    """Raise RuntimeError if a forward on `device` lost a hand-off inside its persistent kernel (its poses are NaN).  Forwards
    are asynchronous, so call this after the LAST batch of a loop (synchronize=True waits for the device first): every other
    batch is covered by the next call into the library, the last one only by this."""
    from . import cabi
    cabi.raise_if_device_error(device, synchronize)
