"""GPU idle time in a rocprofv3 --kernel-trace CSV of bench.py: union of the kernel intervals over the last steps against the wall span,
and which kernels the device waited in front of.  python tools/trace_gaps.py <kernel_trace.csv> [first_step=10] [last_step=20]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# window: the launches between two occurrences of the optimiser kernel (flat_adam) - whole training steps out of the timed region
marks = [i for i, r in enumerate(rows) if "adam" in r[2].lower()]
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 10
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 20
if len(marks) > hi:
    rows = rows[marks[lo] + 1:marks[hi] + 1]
    print("steps %d..%d of the trace: %.3f ms per step" % (lo, hi, (rows[-1][1] - rows[0][0]) / 1e6 / (hi - lo)))
t0 = rows[0][0]
busy, cur_end, gaps = 0, t0, defaultdict(lambda: [0, 0])
overlap = 0
for s, e, name in rows:
    if s > cur_end:
        g = s - cur_end
        short = name.split("(")[0][-60:]
        gaps[short][0] += g; gaps[short][1] += 1
        busy += e - s
        cur_end = e
    else:
        overlap += min(e, cur_end) - s
        if e > cur_end:
            busy += e - cur_end
            cur_end = e
wall = cur_end - t0
print("kernels %d  wall %.2f ms  busy (union) %.2f ms  idle %.2f ms (%.1f %%)  overlapped kernel time %.2f ms" %
      (len(rows), wall / 1e6, busy / 1e6, (wall - busy) / 1e6, 100.0 * (wall - busy) / wall, overlap / 1e6))
hist = defaultdict(int)
prev_end = t0
for s, e, name in rows:
    if s > prev_end:
        g = (s - prev_end) / 1e3
        hist["<2us" if g < 2 else "<5us" if g < 5 else "<10us" if g < 10 else "<50us" if g < 50 else ">=50us"] += 1
    prev_end = max(prev_end, e)
print("gap histogram:", dict(hist))
print("idle in front of (top 25):")
for name, (t, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print("  %8.3f ms  %5d x  %6.2f us  %s" % (t / 1e6, n, t / n / 1e3, name))
# launches per step that are not this library's kernels (torch-native elementwise / fills / cats, runtime blit copies)
nst = max(hi - lo, 1) if len(marks) > hi else 1
other = defaultdict(lambda: [0, 0])
for s, e, name in rows:
    if "at::" in name or "rocclr" in name:
        k = name.split("<")[0] + ("<" + name.split("<")[-1][:60] if "<" in name else "")
        other[k[:110]][0] += 1; other[k[:110]][1] += e - s
print("torch / runtime launches per step (count, us per step):")
for k, (n, t) in sorted(other.items(), key=lambda kv: -kv[1][1])[:30]:
    print("  %6.1f x  %7.1f us  %s" % (n / nst, t / 1e3 / nst, k))
# per HIP stream: busy time and launch count per step (which stream is the critical path, and how full it is)
try:
    per = defaultdict(lambda: [0, 0, None, 0])
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if s < rows[0][0] or e > rows[-1][1] + 1:
                continue
            q = r.get("Stream_Id") or r.get("Queue_Id")
            p = per[q]
            p[0] += e - s; p[1] += 1
            if p[2] is not None and s > p[2]:
                p[3] += s - p[2]
            p[2] = e if p[2] is None else max(p[2], e)
    # the busiest stream's own gaps: which launch did it wait in front of (host behind, or a wait on the other stream)?
    main_q = max(per.items(), key=lambda kv: kv[1][0])[0]
    mg, prev = defaultdict(lambda: [0, 0]), None
    with open(sys.argv[1]) as f:
        rr = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(f)
                    if (r.get("Stream_Id") or r.get("Queue_Id")) == main_q)
    for s_, e_, n_ in rr:
        if s_ < rows[0][0] or e_ > rows[-1][1] + 1:
            continue
        if prev is not None and s_ > prev:
            k = n_.replace("(anonymous namespace)::", "").split("(")[0][-70:]
            mg[k][0] += s_ - prev; mg[k][1] += 1
        prev = e_ if prev is None else max(prev, e_)
    print("stream %s waited in front of (top 12, ms / step):" % main_q)
    for k, (t, n) in sorted(mg.items(), key=lambda kv: -kv[1][0])[:12]:
        print("  %7.3f ms  %5.1f x  %6.2f us  %s" % (t / 1e6 / nst, n / nst, t / n / 1e3, k))
    print("per stream (busy ms / step, launches / step, gaps between consecutive launches of the stream ms / step):")
    for q, (b, n, _, g) in sorted(per.items(), key=lambda kv: -kv[1][0]):
        print("  stream %s: %.3f ms  %.1f launches  gaps %.3f ms" % (q, b / 1e6 / nst, n / nst, g / 1e6 / nst))
except Exception as ex:
    print("per-stream split unavailable:", ex)
