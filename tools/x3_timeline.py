"""Phase timeline INSIDE one launch of the f32x3 3x3 kernel (BIHOME_TUNING=1 build): every workgroup tile stamps the 100 MHz
wall clock at its start, after the first halo stage is in LDS, at the end of the tap loop and after its epilogue.
Prints per-phase durations and how the tiles of one CU interleave.  Usage: BIHOME_TUNING=1 python tools/x3_timeline.py [dgrad]"""
import ctypes, sys; sys.path.insert(0, '.')
import numpy as np, torch
from bihome_amd import kernels as K
from bihome_amd._lib import lib
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
N, H, Ci, Co = 128, 32, 64, 64
PREC = int(__import__("os").environ.get("X3_PREC", "4"))
d = K.conv_desc(N, H, H, Ci, Co, 3, 1, 1, precision=PREC)
x = torch.randn(N, H, H, Ci, device='cuda'); gy = torch.randn(N, H, H, Co, device='cuda')
w = (torch.randn(Co, Ci, 3, 3, device='cuda') * 0.05).contiguous(memory_format=torch.channels_last)
wk = w.permute(0, 2, 3, 1)
pk = K.packer_for_precision(PREC); pf, pd = pk.get(w); pk.refresh()
s = K.bn_stats_buffer(2, Co, "cuda")
def run():
    if mode == "fwd": K.conv_fwd(x, wk, None, d, wpacked=pf)
    elif mode == "fwdstats": K.conv_fwd(x, wk, None, d, bn_sums=s, groups=2, wpacked=pf)
    else: K.conv_dgrad(gy, wk, d, wpacked=pd)
for _ in range(5): run()
torch.cuda.synchronize()
lib.bh_debug_force_tile(-40, 1)
run(); torch.cuda.synchronize()
lib.bh_debug_force_tile(-40, 0)
ntile = N * (H // 8) ** 2 // 2
buf = np.zeros((ntile, 8), np.uint64)
lib.bh_debug_read_c3_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.bh_debug_read_c3_stamps(buf.ctypes.data_as(ctypes.c_void_p), ntile); assert rc == 0, rc
t = buf[:, :4].astype(np.int64); t0 = t[:, 0].min()
t = (t - t0) * 0.01                                        # us
print(mode, "tiles", ntile, "launch span %.1f us" % t[:, 3].max())
for k, name in enumerate(["prologue (start -> first stage in LDS)", "tap loop", "epilogue (incl. store drain)"]):
    dur = t[:, k + 1] - t[:, k]
    print("  %-42s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f us" % (name, dur.mean(), *np.percentile(dur, [10, 50, 90])))
print("  tile start times: p0 %.1f p25 %.1f p50 %.1f p75 %.1f p100 %.1f" % tuple(np.percentile(t[:, 0], [0, 25, 50, 75, 100])))
print("  tile end times:   p0 %.1f p25 %.1f p50 %.1f p75 %.1f p100 %.1f" % tuple(np.percentile(t[:, 3], [0, 25, 50, 75, 100])))
cyc = (buf[:, 6].astype(np.int64) - buf[:, 5].astype(np.int64)); wall = t[:, 2] - t[:, 1]
mhz = cyc / np.maximum(wall, 1e-9)
print("  s_memtime ticks per us of wall clock inside the tap loop: p10 %.0f p50 %.0f p90 %.0f  (ticks per loop p50 %.0f)" % (*np.percentile(mhz, [10, 50, 90]), np.median(cyc)))
hw = buf[:, 4]; xcc = (hw >> np.uint64(32)).astype(np.int64) & 15; hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
cu = (hwid >> 8) & 15; se = (hwid >> 13) & 7; key = xcc * 1000 + se * 16 + cu
print("  distinct (xcc, se, cu):", len(set(key.tolist())))
k0 = key[0]
rows = sorted((t[i, 0], t[i, 1], t[i, 2], t[i, 3], i) for i in range(ntile) if key[i] == k0)
print("  tiles on the CU of tile 0 (start, stage-in, loop end, end):")
for r in rows: print("    tile %4d: %6.1f %6.1f %6.1f %6.1f" % (r[4], r[0], r[1], r[2], r[3]))
# chip-wide activity: fraction of tiles inside their tap loop per microsecond
T = int(t[:, 3].max()) + 1
act = [(int(((t[:, 1] <= u) & (t[:, 2] > u)).sum()), int(((t[:, 2] <= u) & (t[:, 3] > u)).sum()), int(((t[:, 0] <= u) & (t[:, 1] > u)).sum())) for u in range(T)]
print("  per us: tiles in (prologue, loop, epilogue)")
print("   ", " ".join("%d/%d/%d" % (p, l, e) for l, e, p in act))
