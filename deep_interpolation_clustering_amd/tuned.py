"""Pre-tuned library GEMM selections for the dense layers of the joint step.

The GEMMs the hand-written kernels do not cover (the decoder LSTM's dX; at batch sizes / shapes below the kernels' thresholds
also the input projection, the split-K weight gradients and the CompressFC layers) go to hipBLASLt / rocBLAS through PyTorch.
Their default heuristics are off for several of the step's shapes (K = 256 with a 786 432-row output; 184-batch 4096-row split-K
products): PyTorch's TunableOp picks the fastest solution per shape (3-4 % of the round-1 step at B = 32 768).  ``tuned_gemm_gfx950.csv`` holds the result of
``scripts/tune_gemms.py`` on an MI355X (ROCm 7.x validators inside the file: it is ignored on any other stack);
``enable()`` switches TunableOp on in read-only mode with that table.  Shapes not in the table use the default.
"""
import os

import torch

TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tuned_gemm_gfx950.csv')


def enable(table=TABLE):
    """Use the shipped selections (no tuning, nothing written).  Returns True when the table was accepted."""
    if not (torch.cuda.is_available() and os.path.exists(table)):
        return False
    import torch.cuda.tunable as T
    T.enable(True)
    T.tuning_enable(False)
    try:
        return bool(T.read_file(table))
    except Exception:                                   # a table from another ROCm / PyTorch build: keep the defaults
        T.enable(False)
        return False
