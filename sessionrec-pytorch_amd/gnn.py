"""SRGNNLayer on the HIP path (srgnn.py:11-51 / niser.py:11-49): weighted-mean aggregation over the
session graph and its reverse, W1 / W2 projections, GRUCell(cat[W1 n1, W2 n2], feat).

The reference executes these layers but never consumes their output; the product
runs them only under `use_gnn_output=True`."""
from . import ops


def srgnn_layer(layer, mg, feat):
    if mg.count('E') == 0:
        return feat
    ft = layer.dropout(feat)
    N = feat.shape[0]
    in_csr = (mg.in_ptr, mg.in_idx, mg.esrc)
    out_csr = (mg.out_ptr, mg.out_idx, mg.edst)
    c_in = ops.edge_coef(mg.in_ptr, mg.in_idx, mg.ew, N)          # w_e / sum of w into dst(e)
    c_out = ops.edge_coef(mg.out_ptr, mg.out_idx, mg.ew, N)       # reversed graph: w_e / sum of w out of src(e)
    neigh1 = ops.edge_agg(ft, c_in, in_csr, out_csr)
    neigh2 = ops.edge_agg(ft, c_out, out_csr, in_csr)
    a = ops.linear(neigh1, layer.W1.weight)
    b = ops.linear(neigh2, layer.W2.weight)
    gi = ops.linear_cat([a, b], layer.gru.weight_ih, layer.gru.bias_ih)
    gh = ops.linear(feat, layer.gru.weight_hh, layer.gru.bias_hh)
    return ops.gru_step(gi, gh, None, feat)
