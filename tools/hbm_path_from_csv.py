"""The warp + perceptual path (BASELINE.json's HBM-bound part) from rocprofv3 kernel_stats.csv files: sum of the average durations of the
path's launches of a step and the fraction of 8 TB/s for the 205 MB they move.
    python tools/hbm_path_from_csv.py profiles/r06x_kernel_stats.csv [profiles/r06x_kernel_stats_unfolded.csv]
Round 6: the warp's adjoint runs inside the extractor stem's dgrad (stem7_dgrad_c1_kernel<true>), there is no warp_bwd4_kernel launch; its
cost is what it adds to that launch - the fused kernel's average duration minus the plain stem7_dgrad_c1_kernel<false>'s, taken from the
second csv (the same step with BIHOME_WARP_IN_STEM_DGRAD=0 BIHOME_WARP_IN_STEM_FWD=0, where warp_bwd4_kernel also shows what the separate
launch cost).  The forward warp likewise runs inside the extractor stem's forward (stem7_fwd_f16_kernel<1, true>): no warp_fwd4_kernel
launch; its cost is that kernel's average duration minus the plain stem7_fwd_f16_kernel<1, false>'s of the SAME trace (the extractor runs
once per step on the unwarped patches: same shape, same stream in a --no-overlap trace)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else "profiles/r04j_kernel_stats.csv")))
rows2 = list(csv.DictReader(open(sys.argv[2]))) if len(sys.argv) > 2 else []
def avg(rs, name):
    r = [x for x in rs if name in x["Name"]]
    return (float(r[0]["AverageNs"]) / 1e3, r[0]["Calls"]) if r else (float("nan"), "-")
tot = 0.0
for w in ("triplet_fwd_kernel", "triplet_bwd_kernel"):
    us, calls = avg(rows, w)
    print("%-46s %7.2f us (%s calls)" % (w, us, calls))
    tot += us
wf, calls = avg(rows, "warp_fwd4_kernel")
if wf == wf:
    print("%-46s %7.2f us (%s calls)" % ("warp_fwd4_kernel", wf, calls))
    tot += wf
else:
    fused, c1 = avg(rows, "stem7_fwd_f16_kernel<1, true")
    plain, c2 = avg(rows, "stem7_fwd_f16_kernel<1, false")
    sep, c3 = avg(rows2, "warp_fwd4_kernel")
    print("%-46s %7.2f us (%s calls)" % ("stem7_fwd_f16_kernel<1,true> (warp inside)", fused, c1))
    print("%-46s %7.2f us (%s calls; same csv)" % ("stem7_fwd_f16_kernel<1,false>", plain, c2))
    print("%-46s %7.2f us (%s calls; second csv: the launch the fold removed)" % ("warp_fwd4_kernel", sep, c3))
    add = max(fused - plain, 0.0)
    print("%-46s %7.2f us" % ("warp = fused - plain", add))
    tot += add
wb, calls = avg(rows, "warp_bwd4_kernel")
if wb == wb:
    print("%-46s %7.2f us (%s calls)" % ("warp_bwd4_kernel", wb, calls))
    tot += wb
else:
    fused, c1 = avg(rows, "stem7_dgrad_c1_kernel<true")
    plain, c2 = avg(rows2, "stem7_dgrad_c1_kernel<false")
    sep, c3 = avg(rows2, "warp_bwd4_kernel")
    print("%-46s %7.2f us (%s calls)" % ("stem7_dgrad_c1_kernel<true> (warp adjoint inside)", fused, c1))
    print("%-46s %7.2f us (%s calls; second csv)" % ("stem7_dgrad_c1_kernel<false>", plain, c2))
    print("%-46s %7.2f us (%s calls; second csv: the launch the fold removed)" % ("warp_bwd4_kernel", sep, c3))
    add = max(fused - plain, 0.0)
    print("%-46s %7.2f us" % ("warp adjoint = fused - plain", add))
    tot += add
nbytes = 204996608.0
tbs = nbytes / (tot * 1e-6) / 1e12
print("sum %.1f us for %.0f MB -> %.2f TB/s = %.3f of 8 TB/s" % (tot, nbytes / 1e6, tbs, tbs / 8.0))
