"""Config C1 (BASELINE.json configs[0]): SRGNN on the reference's shipped datasets/sample split, batch 32,
driven by the TrainRunner exactly like start.sh does - the HIP path on the GPU must reproduce the loss trace
and the (MRR@20, HR@20) of the CPU oracle on the same batches."""
import os

import numpy as np
import pytest
import torch

from util import ROOT, pkg

pytestmark = pytest.mark.gpu


class _Wrap:
    def __init__(self, x):
        self.x = x

    def to(self, device):
        return self.x


@pytest.mark.parametrize('model_name', ['SRGNN', 'LESSR', 'MSGIFSR'])
def test_sample_dataset_training_matches_oracle(dev, model_name):
    from oracle import collate_ref as oc, models_ref as om
    sp, ds, col, train = pkg(), pkg('dataset'), pkg('collate'), pkg('train')
    tr, te, V = ds.read_dataset(os.path.join(ROOT, 'datasets', 'sample'))
    train_set, test_set = ds.AugmentedDataset(tr), ds.AugmentedDataset(te)
    B, n_train, n_test = 32, 12, 6
    torch.manual_seed(123)
    if model_name == 'SRGNN':
        ref, model = om.SRGNN(V, 32, 1), sp.SRGNN(V, 32, 1)
        cf, of = col.collate_fn_factory(col.seq_to_session_graph), oc.collate_fn_factory(oc.seq_to_session_graph)
    elif model_name == 'LESSR':
        ref, model = om.LESSR(V, 32, 2), sp.LESSR(V, 32, 2)
        cf = col.collate_fn_factory(col.seq_to_eop_multigraph, col.seq_to_shortcut_graph)
        of = oc.collate_fn_factory(oc.seq_to_eop_multigraph, oc.seq_to_shortcut_graph)
    else:
        ref = om.MSGIFSR(V, 'sample', 32, 1, order=2, extra=False, fusion=False)
        model = sp.MSGIFSR(V, 'sample', 32, 1, order=2, extra=False, fusion=False)
        cf = col.collate_fn_factory_ccs((col.seq_to_ccs_graph,), 2)
        of = oc.collate_fn_factory_ccs((oc.seq_to_ccs_graph,), 2)
    model.load_state_dict(ref.state_dict())
    model = model.to(dev)

    def batches(data, n, fn, wrap):
        out = []
        for b in range(n):
            inp, lab = fn([data[i] for i in range(b * B, (b + 1) * B)])
            if wrap:
                out.append(([_Wrap(om.to_torch(x)) for x in inp], torch.from_numpy(lab)))
            else:
                out.append((inp, lab))
        return out
    r_ref = train.TrainRunner('sample', ref, batches(train_set, n_train, of, True), batches(test_set, n_test, of, True),
                              torch.device('cpu'), lr=1e-3, weight_decay=1e-4, patience=2)
    r_gpu = train.TrainRunner('sample', model, batches(train_set, n_train, cf, False), batches(test_set, n_test, cf, False),
                              dev, lr=1e-3, weight_decay=1e-4, patience=2)
    assert r_gpu.fused
    m_ref = r_ref.train(2, log_interval=100)
    m_gpu = r_gpu.train(2, log_interval=100)
    a, b = np.array(r_gpu.loss_trace), np.array(r_ref.loss_trace)
    assert a.shape == b.shape == (2 * n_train,)
    assert np.allclose(a, b, rtol=2e-4, atol=2e-4), np.abs(a - b).max()
    assert abs(m_gpu[0] - m_ref[0]) <= 0.02 and abs(m_gpu[1] - m_ref[1]) <= 0.02, (m_gpu, m_ref)   # (MRR@20, HR@20)
