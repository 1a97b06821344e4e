"""`from src.utils.train import TrainRunner` (reference: src/utils/train.py) -> sessionrec-pytorch_amd.train"""
import src.models  # noqa: F401  (puts the repo root on sys.path)
from importlib import import_module as _im

_m = _im('sessionrec-pytorch_amd.train')
TrainRunner, evaluate, fix_weight_decay, prepare_batch = _m.TrainRunner, _m.evaluate, _m.fix_weight_decay, _m.prepare_batch
