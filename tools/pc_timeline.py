"""Round 5: barrier time line of ONE workgroup of conv3x3_pc_kernel (BH_TUNING build): for each role (C consumers, H halo staging, D weight
DMA, E epilogue) the shader-clock time at which its first wave arrives at / leaves every barrier.  Shows who the others wait for.
BIHOME_TUNING=1 python tools/pc_timeline.py [N,H,Ci,Co] [mode: fwd|stats|bnr]"""
import sys, ctypes; sys.path.insert(0, '.')
import numpy as np, torch
from bihome_amd import kernels as K
from bihome_amd._lib import lib, ROUTE_C3_PC
shape = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (128, 32, 64, 64)
mode = sys.argv[2] if len(sys.argv) > 2 else "fwd"
N, H, Ci, Co = shape
x = torch.randn(N, H, H, Ci, device='cuda'); gy = torch.randn(N, H, H, Co, device='cuda')
w = (torch.randn(Co, Ci, 3, 3, device='cuda') * 0.05).contiguous(memory_format=torch.channels_last)
wk = w.permute(0, 2, 3, 1)
pk = K.packer_for_precision(4); pf, pd = pk.get(w); pk.refresh()
z = torch.randn(N, H, H, Ci, device='cuda') * 1.5 + 0.3
gamma, beta = torch.rand(Ci, device='cuda') + 0.5, torch.randn(Ci, device='cuda') * 0.2
st = K.bn_stats_buffer(2, Ci, "cuda"); K.bn_stats(z, st, 2, Ci)
base = torch.randn(N, H, H, Ci, device='cuda')
d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=4, route=ROUTE_C3_PC)
s = K.bn_stats_buffer(2, Co, "cuda"); s2 = K.bn_stats_buffer(2, Ci, "cuda")
bnr = dict(z=z, y=None, stats=st, gamma=gamma, beta=beta, eps=1e-5, relu=True, sums=s2, groups=2)
fn = {"fwd": lambda: K.conv_fwd(x, wk, None, d, wpacked=pf), "stats": lambda: K.conv_fwd(x, wk, None, d, bn_sums=s, groups=2, wpacked=pf),
      "bnr": lambda: K.conv_dgrad(gy, wk, d, out=base, wpacked=pd, bn_reduce=bnr)}[mode]
for _ in range(5): fn()
torch.cuda.synchronize()
lib.bh_debug_force_tile(-40, 1 + 37)          # stamps of workgroup 37
for _ in range(3): fn()
torch.cuda.synchronize()
lib.bh_debug_force_tile(-40, 0)
buf = (ctypes.c_ulonglong * (12 * 160 * 2))()
lib.bh_debug_read_pc_stamps.argtypes = [ctypes.c_void_p]
lib.bh_debug_read_pc_stamps(buf)
tw = np.array(buf, dtype=np.float64).reshape(12, 160, 2)          # round 6: every wave (0-3 C, 4-6 H, 7 D, 8-11 E)
nb = int((tw[0, :, 0] > 0).sum())
t0 = tw[:, 0, 0].min()
tw = (tw - t0) / 1000.0          # kilo-cycles
ROLE = "CCCCHHHDEEEE"
rep = [0, 4, 7, 8]               # one wave per role for the table (as in round 5)
t = tw[rep]
print(shape, mode, "barriers", nb, " (kilo-cycles of the shader clock; arrival -> release per role; last = the wave the others waited for, its lag behind the second-last)")
print(" b#    C arr   rel |   H arr   rel |   D arr   rel |   E arr   rel |  last wave")
for b in range(nb):
    arr = t[:, b, 0]; rel = t[:, b, 1]
    a12 = tw[:, b, 0]; order = np.argsort(a12); w = int(order[-1])
    print("%3d  " % b + " | ".join("%7.2f %6.2f" % (arr[r], rel[r] - arr[r]) for r in range(4)) + " |  %s%d +%.2f" % (ROLE[w], w, a12[order[-1]] - a12[order[-2]]))
print("total %.1f kilo-cycles; waiting at barriers: C %.1f  H %.1f  D %.1f  E %.1f" % ((tw[:, nb - 1, 1].max(),) + tuple((t[r, :nb, 1] - t[r, :nb, 0]).sum() for r in range(4))))
last = [ROLE[int(tw[:, b, 0].argmax())] for b in range(nb)]
print("last to arrive: " + "  ".join("%s %d" % (r, last.count(r)) for r in "CHDE"))
lagw = {}
for b in range(nb):
    a12 = tw[:, b, 0]; order = np.argsort(a12); w = int(order[-1])
    lagw[w] = lagw.get(w, 0.0) + (a12[order[-1]] - a12[order[-2]])
print("kilo-cycles the workgroup waited for its last wave, by wave: " + "  ".join("%s%d %.1f" % (ROLE[w], w, v) for w, v in sorted(lagw.items())))
