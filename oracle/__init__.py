"""oracle/ -- CPU restatements used ONLY as checkers (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).

mpl_oracle.py      the reference's forward pass (MPL/lib/models/multiview_mpl.py), pinned on goldens captured from the
                   reference itself (tests/golden/, generated through ref_import.py)
metrics_oracle.py  validate()'s host epilogue (function_mpl.py, evaluate.py, loss.py)
inputs_oracle.py   the dataset's per-sample input preparation (joints_dataset_mpl.py)
split_oracle.py    the definition of the build's own split weight operand (not a reference restatement)

Nothing under openmpl_amd/ imports this package: the product path has no CPU fallback.
"""
