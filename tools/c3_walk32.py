"""Round 4: 32-channel fp16-piece 3x3 launches without statistics walking several tile positions per workgroup - on (default) against off
(knob -43, 0): same bits; time per launch, the variants alternating (the first of two back-to-back measurements is slower whatever it is).
BIHOME_TUNING=1 python tools/c3_walk32.py"""
import sys; sys.path.insert(0, '.')
import torch
from bihome_amd import kernels as K
from bihome_amd._lib import lib
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
torch.manual_seed(0)
for (N, H, Ci, Co, groups) in [(128, 128, 32, 32, 2), (128, 64, 32, 32, 2), (128, 64, 64, 32, 1), (64, 128, 32, 32, 1)]:
    d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=4)
    x = torch.relu(torch.randn(N, H, H, Ci, device='cuda'))
    gy = torch.randn(N, H, H, Co, device='cuda') * (torch.rand(N, H, H, Co, device='cuda') > 0.5)
    w = (torch.randn(Co, Ci, 3, 3, device='cuda') * 0.05).contiguous(memory_format=torch.channels_last)
    wk = w.permute(0, 2, 3, 1)
    pk = K.packer_for_precision(4); pf, pd = pk.get(w); pk.refresh()
    sums = K.bn_stats_buffer(groups, Co, 'cuda')
    def fwd_stats():
        sums.zero_()
        return K.conv_fwd(x, wk, None, d, bn_sums=sums, groups=groups, wpacked=pf)
    for name, fn in (("fwd", lambda: K.conv_fwd(x, wk, None, d, wpacked=pf)), ("fwd+stats", fwd_stats), ("dgrad", lambda: K.conv_dgrad(gy, wk, d, wpacked=pd))):
        res, tm = {}, {0: [], 1: []}
        for on in (0, 1, 0, 1, 0, 1):
            lib.bh_debug_force_tile(-43, on)
            try:
                y = fn()
            except Exception as e:
                print((N, H, Ci, Co), name, 'unsupported'); break
            y = y[0] if isinstance(y, tuple) else y
            res[on] = (y.clone(), sums.clone())
            tm[on].append(bench(fn))
        else:
            same = torch.equal(res[0][0], res[1][0])
            ssame = torch.allclose(res[0][1], res[1][1], rtol=1e-9, atol=1e-6) if name == "fwd+stats" else True
            print((N, H, Ci, Co, groups), name, 'off', ' '.join('%.1f' % t for t in tm[0]), ' on', ' '.join('%.1f' % t for t in tm[1]), ' bits equal:', same, ssame, flush=True)
            assert same and ssame
lib.bh_debug_force_tile(-43, 1)
