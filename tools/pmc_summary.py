#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, kernel-trace only) into a per-kernel
HBM-traffic summary.  Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM section): the counters are
in KiB; FETCH_SIZE under-reports wide coalesced (16 B/lane) streaming reads by exactly 2x on this rocprofv3, which
is re-checked here on this build's own pure streaming kernel (bn_apply: reads == ~1.3x writes by construction).

    python tools/pmc_summary.py gpurun_out/pmc_r01_FETCH_SIZE/pmc_counter_collection.csv \
                                gpurun_out/pmc_r01_WRITE_SIZE/pmc_counter_collection.csv profiles/r01_pmc_traffic.json
"""
import collections
import csv
import json
import re
import sys


def agg(path):
    out = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "")
        k = re.sub(r"\(.*$", "", k).replace(" ", "")
        out[k][0] += 1
        out[k][1] += float(r["Counter_Value"])
    return out


def main():
    f, w = agg(sys.argv[1]), agg(sys.argv[2])
    res = {}
    for k in f:
        n = f[k][0]
        fk, wk = f[k][1] / n, (w[k][1] / w[k][0] if k in w else 0.0)
        res[k] = {"launches": n, "fetch_kib_avg_raw": fk, "write_kib_avg": wk,
                  "traffic_bytes_per_launch": (2.0 * fk + wk) * 1024.0}
    cal = res.get("bn_apply_kernel")
    meta = {"correction": "traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes (gfx950 FETCH_SIZE halves 16 B/lane streams)",
            "calibration_bn_apply_fetch_over_write_raw": (cal["fetch_kib_avg_raw"] / cal["write_kib_avg"]) if cal else None}
    json.dump({"meta": meta, "kernels": res}, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"])[:8]:
        print("%-45s %6d launches  %8.1f MB/launch" % (k, v["launches"], v["traffic_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
