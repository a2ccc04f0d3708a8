"""Round 6: the fp16-piece weight gradient with the BatchNorm adjoint applied on load (bh_conv_wgrad_bnadj) against the plain launch on the
materialised adjoint, alone (kernel + reduce per call), four- and eight-wave forms, on the step's three shapes.  python tools/wx3_bnadj_ab.py"""
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from bihome_amd import kernels as K
from bihome_amd._lib import ROUTE_WX3_PC, ROUTE_WX3_SHARED
def bench(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n
for (N, H, C) in [(128, 32, 64), (128, 16, 128), (128, 8, 256)]:
    groups = 2
    x = torch.randn(N, H, H, C, device='cuda'); z = torch.randn(N, H, H, C, device='cuda') * 2 + 0.5
    dout = torch.randn(N, H, H, C, device='cuda') * 1e-3
    gamma, beta = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.2
    rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    y, st = K.bn_fwd(z, gamma, beta, rm, rv, None, groups, 1e-5, 0.1, True, True)
    sums = K.bn_stats_buffer(groups, C, 'cuda')
    sums.view(-1)[::16] = torch.randn(sums.numel() // 16, dtype=torch.float64, device='cuda')        # (some backward sums: timing only)
    rec_d = K.amax_record('cuda'); rec_d[0] = float(dout.abs().max())
    recg = K.amax_record('cuda')
    gx, _ = K.bn_bwd(dout, None, z, gamma, st, rm, rv, groups, 1e-5, True, True, False, beta=beta, had_res=False, sums_ready=sums, amax=recg)
    K.amax_of(x)
    out = []
    for name, route in (("four waves", 0), ("four waves, 160 workgroups", ROUTE_WX3_SHARED), ("eight waves", ROUTE_WX3_PC)):
        d = K.conv_desc(N, H, H, C, C, 3, 1, 1, precision=4, route=route); d.bh_wx3 = True
        ws = torch.empty(K.wgrad_det_bytes(d) // 4, dtype=torch.float32, device='cuda')
        gw = torch.zeros(C, 3, 3, C, device='cuda')
        bna = dict(z=z, y=None, stats=st, sums=sums, gamma=gamma, beta=beta, eps=1e-5, relu=True, groups=groups)
        bnay = dict(bna, y=y)
        t0 = bench(lambda: K.conv_wgrad(x, gx, gw, None, d, det_ws=ws))
        t1 = bench(lambda: K.conv_wgrad_bnadj(x, dout, gw, d, ws, bna, rec_d))
        t2 = bench(lambda: K.conv_wgrad_bnadj(x, dout, gw, d, ws, bnay, rec_d))
        out.append("%s: plain %.1f | adjoint on load %.1f | ... with the saved output %.1f us" % (name, t0, t1, t2))
    tb = bench(lambda: K.bn_bwd(dout, None, z, gamma, st, rm, rv, groups, 1e-5, True, True, False, beta=beta, had_res=False, sums_ready=sums, amax=recg))
    print((N, H, C), " || ".join(out), "|| the adjoint pass alone %.1f us" % tb, flush=True)
