"""Development probe: what the fused scoring/CE backward (flash_ce_bf16_kernel<8, 1>, B 512, V 37 484, d 256) would take with
one of its costs removed - private builds of score_ce_bf16.hip with -DSREC_FLASH_KO=<bits> (bit 0: no exp, 1: no accumulate
product, 2: no S product, 3: no result stores, 4: MFMA fragments from registers instead of LDS, 5: the logits read back from
memory as fp16 fragments instead of recomputed - what a forward that STORES them would buy the backward), each timed in its own
process (HIP events over 20 launches of the merged backward).  Results of a knocked-out build are garbage by construction.
usage (GPU box): python tools/flash_knockout.py [bits ...]"""
import glob, importlib, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pk = os.path.join(root, 'sessionrec-pytorch_amd')


def child(ko):
    sys.path.insert(0, root)
    L = importlib.import_module('sessionrec-pytorch_amd._lib')
    L.LIB_PATH = '/tmp/libsrec_ko%d.so' % ko
    import torch
    import bench
    ops = importlib.import_module('sessionrec-pytorch_amd.ops')
    sp = importlib.import_module('sessionrec-pytorch_amd')
    ops.set_precision('bf16')
    dev = torch.device('cuda:0')
    B, V, d = 512, 37484, 256
    torch.manual_seed(123)
    model = bench.build_model(sp, 'SRGNN', V, d, 1).to(dev)
    table = model._table().detach()[:V]
    sr = torch.randn(B, d, device=dev) * 0.1
    labels = torch.randint(0, V, (B,), device=dev, dtype=torch.int32)
    ws = ops.CEWorkspace(B, V, d, dev)
    lse, lossvec, loss = torch.empty(B, device=dev), torch.empty(B, device=dev), torch.empty((), device=dev)
    dE, dsr = torch.empty_like(table), torch.empty(B, d, device=dev)
    tb = ops.TableBF16(table).refresh(table)
    ops._ce_fwd(sr, table, None, labels, ws, None, tb, ws.lab_logit, lse, lossvec, loss)
    for _ in range(5):
        ops._ce_bwd(sr, table, None, labels, lse, None, None, None, ws, None, tb, dE, dsr, 3)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops._ce_bwd(sr, table, None, labels, lse, None, None, None, ws, None, tb, dE, dsr, 3)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 20)
    print('KO %2d: %.1f us per backward (launch + d-sr slab sum; min of 5 x 20: %.1f)' % (ko, sorted(ts)[2], min(ts)), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--child':
        child(int(sys.argv[2]))
        sys.exit(0)
    kos = [int(a) for a in sys.argv[1:]] or [0, 1, 16, 17, 8, 25, 2, 4, 27, 29]
    objs = [o for o in glob.glob(pk + '/csrc/*.o') if not o.endswith('score_ce_bf16.o')]
    names = {1: 'no exp', 2: 'no accumulate product', 4: 'no S product', 8: 'no stores', 16: 'no LDS fragment reads',
             32: 'logits read back from memory (fp16 fragments, one chunk ahead) instead of recomputed'}
    for ko in kos:
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-DSREC_FLASH_KO=%d' % ko,
                               '-I', root + '/include', '-c', pk + '/csrc/score_ce_bf16.hip', '-o', '/tmp/score_ce_bf16_ko.o'])
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', '/tmp/libsrec_ko%d.so' % ko,
                               '/tmp/score_ce_bf16_ko.o'] + objs)
        print('   (%s)' % (', '.join(v for k, v in names.items() if ko & k) or 'the product kernel'), flush=True)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), '--child', str(ko)])
