"""For every launch whose name contains PATTERN in a rocprofv3 kernel trace csv: the kernels right before and after it (which code path
issues it?).  Usage: python tools/trace_neighbors.py kernel_trace.csv copyBuffer"""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2]
cnt = collections.Counter()
for i, r in enumerate(rows):
    if pat in r["Kernel_Name"]:
        prev = rows[i - 1]["Kernel_Name"][:60] if i else "-"
        nxt = rows[i + 1]["Kernel_Name"][:60] if i + 1 < len(rows) else "-"
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        cnt[(prev, nxt)] += 1
for (p, n), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print("%4d  after [%s]  before [%s]" % (c, p, n))
