"""The warp + perceptual path (BASELINE.json's HBM-bound part) from a rocprofv3 kernel_stats.csv: sum of the average durations of the four
launches of a step and the fraction of 8 TB/s for the 205 MB they move.  python tools/hbm_path_from_csv.py profiles/r04j_kernel_stats.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else "profiles/r04j_kernel_stats.csv")))
want = ("triplet_fwd_kernel", "triplet_bwd_kernel", "warp_fwd4_kernel", "warp_bwd4_kernel")
tot = 0.0
for w in want:
    r = [x for x in rows if w in x["Name"]]
    us = float(r[0]["AverageNs"]) / 1e3 if r else float("nan")
    print("%-20s %7.2f us (%s calls)" % (w, us, r[0]["Calls"] if r else "-"))
    tot += us
nbytes = 204996608.0
tbs = nbytes / (tot * 1e-6) / 1e12
print("sum %.1f us for %.0f MB -> %.2f TB/s = %.3f of 8 TB/s" % (tot, nbytes / 1e6, tbs, tbs / 8.0))
