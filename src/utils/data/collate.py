"""reference surface src/utils/data/collate.py -> sessionrec-pytorch_amd.collate (FlatBatch instead of DGL graphs)"""
import src.models  # noqa: F401
from importlib import import_module as _im

_m = _im('sessionrec-pytorch_amd.collate')
seq_to_eop_multigraph, seq_to_shortcut_graph = _m.seq_to_eop_multigraph, _m.seq_to_shortcut_graph
seq_to_session_graph, seq_to_ccs_graph = _m.seq_to_session_graph, _m.seq_to_ccs_graph
collate_fn_factory, collate_fn_factory_ccs = _m.collate_fn_factory, _m.collate_fn_factory_ccs
estimate_caps, default_caps, measure_caps = _m.estimate_caps, _m.default_caps, _m.measure_caps
