"""Round 4: would running the two directions of the Siamese batch as two half-size launch sequences on two HIP streams hide the fixed cost of
the short 3x3 launches?  One stream x N=128 against two streams x N=64, same layers, alternating.  python tools/split_streams.py"""
import sys; sys.path.insert(0, '.')
import torch
from bihome_amd import kernels as K
torch.manual_seed(0)
def setup(N, H, C, prec=4):
    d = K.conv_desc(N, H, H, C, C, 3, 1, 1, precision=prec, route=2)   # (BH_ROUTE_HALO_SMALL: the 8x8 half batch is 128 workgroups)
    x = torch.relu(torch.randn(N, H, H, C, device='cuda'))
    w = (torch.randn(C, C, 3, 3, device='cuda') * 0.05).contiguous(memory_format=torch.channels_last)
    pk = K.packer_for_precision(prec); pf, pd = pk.get(w); pk.refresh()
    return d, x, w.permute(0, 2, 3, 1), pf, pd, pk
def chain(s, reps):
    d, x, wk, pf, pd, _ = s
    for _ in range(reps):
        K.conv_fwd(x, wk, None, d, wpacked=pf)
        K.conv_dgrad(x, wk, d, wpacked=pd)
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for (H, C) in [(32, 64), (16, 128), (8, 256), (64, 64), (128, 32)]:
    full = setup(128, H, C); ha = setup(64, H, C); hb = setup(64, H, C)
    reps = 20
    def one():
        chain(full, reps)
    def two():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1): chain(ha, reps)
        with torch.cuda.stream(s2): chain(hb, reps)
        cur.wait_stream(s1); cur.wait_stream(s2)
    def halves_serial():
        chain(ha, reps); chain(hb, reps)
    out = []
    for r in range(3):
        out.append((timed(one), timed(two), timed(halves_serial)))
    print((H, C), ' | '.join('one %.3f two-streams %.3f halves-serial %.3f' % o for o in out), 'ms per %d fwd+dgrad' % reps, flush=True)
