import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module('sessionrec-pytorch_amd._lib')
lib, ptr, stream = L.lib, L.ptr, L.stream
dev = torch.device('cuda:0')
for w, B, use_lab in ((1, 512, False), (1, 512, True), (2, 512, True), (8, 4096, True)):
    st = torch.randn(w, 2, B, device=dev)
    lab_all = torch.randint(-1, 5, (B,), device=dev, dtype=torch.int64) if use_lab else None
    lse = torch.empty(B, device=dev); lab = torch.empty(B, device=dev); gw = torch.empty(B, device=dev)
    loss = torch.empty((), device=dev)
    try:
        lib.srec_merge_stats(ptr(st), w, B, ptr(lab_all), ptr(lse), ptr(lab), ptr(loss), ptr(gw), stream())
        torch.cuda.synchronize()
        ref_lse = torch.logsumexp(st[:, 0], 0); ref_lab = st[:, 1].sum(0)
        live = torch.ones(B, device=dev) if lab_all is None else (lab_all >= 0).float()
        ref_loss = ((ref_lse - ref_lab) * live).sum() / live.sum().clamp(min=1)
        print(w, B, use_lab, 'ok', (lse - ref_lse).abs().max().item(), (loss - ref_loss).abs().item(),
              (gw - live / live.sum().clamp(min=1)).abs().max().item())
    except Exception as e:
        print(w, B, use_lab, 'FAILED', e)
