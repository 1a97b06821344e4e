"""Importable alias for the hyphenated package directory `sessionrec-pytorch_amd/`."""
import importlib
import sys

_pkg = importlib.import_module('sessionrec-pytorch_amd')
sys.modules[__name__] = _pkg
