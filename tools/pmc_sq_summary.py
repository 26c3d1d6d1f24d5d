#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc SQ_* passes of bench.py into per-kernel ratios.

  python tools/pmc_sq_summary.py OUT.json DIR1 [DIR2 ...]

Each DIR is the -d directory of one pass.  Counters of different passes are merged per kernel name (averages per wave /
ratios to SQ_WAVE_CYCLES or SQ_BUSY_CYCLES of their own pass, so passes do not have to see identical launch counts).
"""
import collections
import csv
import glob
import json
import re
import sys


def load(d):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        n = re.sub(r"<.*$", "", re.sub(r"^void\s+", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).split("(")[0].replace("gsx::", "")).strip()
        agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
    return agg


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    res = collections.defaultdict(dict)
    for d in dirs:
        for k, c in load(d).items():
            waves = max(c.get("SQ_WAVES", 0.0), 1.0)
            wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
            busy = max(c.get("SQ_BUSY_CYCLES", 0.0), 1.0)
            for name, v in c.items():
                if name in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"):
                    continue
                if name.startswith("SQ_INSTS"):
                    res[k][name + "_per_wave"] = v / waves
                elif name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                              "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"):
                    res[k][name + "_over_wave_cycles"] = v / wc
                else:
                    res[k][name + "_over_busy_cycles"] = v / busy
            if "SQ_WAVE_CYCLES" in c and "SQ_WAVES" in c:
                res[k]["wave_cycles_per_wave"] = c["SQ_WAVE_CYCLES"] / waves
    json.dump({"note": "rocprofv3 --pmc SQ_* of `bench.py --steps 2 --warmup 1 --no-cpu-baseline`; ratios per kernel",
               "kernels": res}, open(out, "w"), indent=1, sort_keys=True)
    for k in ("assemble_h_kernel", "big_gather_seg_kernel", "front_leaf_kernel", "big_panel_kernel", "big_potrf0_kernel",
              "backsolve_kernel", "backsolve_leaf_kernel", "linear_error_kernel"):
        if k in res:
            print(k, {a: round(b, 3) for a, b in sorted(res[k].items())})


if __name__ == "__main__":
    main()
