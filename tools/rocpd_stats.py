"""Per-kernel summary (calls, avg / min / max ns, share) of a rocprofv3 rocpd database: python tools/rocpd_stats.py results.db [filter ...]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
flt = sys.argv[2:]
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for r in rows:
    if flt and not any(f in r[0] for f in flt):
        continue
    print('"%s",%d,%d,%.1f,%.2f,%d,%d' % (r[0], r[1], r[2], r[3], 100.0 * r[2] / tot, r[4], r[5]))
