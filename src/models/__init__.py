"""`from src.models import LESSR, MSGIFSR, NISER, SRGNN` - same import surface as the reference
(/root/reference/src/models/__init__.py:1-4), backed by the MI355X HIP path."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
_pkg = importlib.import_module('sessionrec-pytorch_amd')
LESSR, MSGIFSR, NISER, SRGNN = _pkg.LESSR, _pkg.MSGIFSR, _pkg.NISER, _pkg.SRGNN
