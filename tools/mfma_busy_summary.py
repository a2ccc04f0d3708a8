#!/usr/bin/env python3
"""Per-kernel MFMA-pipe utilisation from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass (kernel-trace
only, the program directly after `--`):

    busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)

(SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs, GRBM_GUI_ACTIVE over its 8 XCDs; units per
MI355X_MICROARCH.md "Per-instruction cycle constants".)

    python tools/mfma_busy_summary.py gpurun_out/pmc_r02a_mfma/r02a_counter_collection.csv profiles/r02a_mfma_busy.json
"""
import collections
import csv
import json
import re
import sys


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(sys.argv[1])):
        k = re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "")
        k = re.sub(r"\(.*$", "", k).replace(" ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[k] += 1
    out = {}
    for k, v in agg.items():
        busy, act = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0)
        if busy > 0 and act > 0:
            out[k] = {"launches": cnt[k], "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": act,
                      "mfma_busy_frac": round(busy / (1024.0 * act / 8.0), 4)}
    meta = {"formula": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8)", "source": sys.argv[1]}
    json.dump({"meta": meta, "kernels": out}, open(sys.argv[2], "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["SQ_VALU_MFMA_BUSY_CYCLES"])[:10]:
        print("%-60s %5d launches  MFMA busy %.3f" % (k[:60], v["launches"], v["mfma_busy_frac"]))


if __name__ == "__main__":
    main()
