"""sessionrec-pytorch_amd: MI355X (gfx950) native training hot path for graph-based session
recommenders (SRGNN / NISER / LESSR / MSGIFSR), drop-in for the model / collate / train-loop
surface of SpaceLearner/SessionRec-pytorch.

The directory name carries a hyphen (fixed by the project layout); import it with
`importlib.import_module('sessionrec-pytorch_amd')` or via the `sessionrec_pytorch_amd` alias
module at the repo root.
"""
from .batch import FlatBatch  # noqa: F401
from .srgnn import NISER, SRGNN  # noqa: F401
from .msgifsr import MSGIFSR  # noqa: F401
from .lessr import LESSR  # noqa: F401
