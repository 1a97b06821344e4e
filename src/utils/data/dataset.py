"""reference surface src/utils/data/dataset.py -> sessionrec-pytorch_amd.dataset"""
import src.models  # noqa: F401
from importlib import import_module as _im

_m = _im('sessionrec-pytorch_amd.dataset')
AugmentedDataset, create_index, read_dataset, read_sessions = _m.AugmentedDataset, _m.create_index, _m.read_dataset, _m.read_sessions
