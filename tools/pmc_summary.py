#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the hardware needs) of bench.py
into per-kernel HBM traffic per launch.

  python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_bal1723_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE under-reports wide (16 B/lane)
streaming reads by 2x (MI355X_MICROARCH.md); this path reads 8 B/lane, for which the guide gives no calibration, so
the raw value is kept and `fetch_calibration` says so.
"""
import collections
import csv
import glob
import json
import re
import subprocess
import sys


def csrc_digest(root="."):
    """sha1 over the device + host sources of libgsx: bench.py reports `traffic` only while it matches the tree it runs from."""
    import hashlib
    import os
    h = hashlib.sha1()
    d = os.path.join(root, "gtsam_petercdev_amd", "csrc")
    for n in sorted(os.listdir(d)):
        if n.endswith((".hip", ".cpp", ".h")):
            h.update(n.encode())
            h.update(open(os.path.join(d, n), "rb").read())
    return h.hexdigest()


def kernel_name(full):
    """'void gsx::(anonymous namespace)::big_diag_kernel<12, 7>(gsx::BigDesc const*, ...)' -> 'big_diag_kernel'"""
    n = full.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void\s+", "", n).split("(")[0].replace("gsx::", "")
    return re.sub(r"<.*$", "", n).strip()


def load(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = kernel_name(r["Kernel_Name"])
        agg[n][0] += 1
        agg[n][1] += float(r["Counter_Value"])
    return agg


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    try:
        sha = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        sha = ""
    res = {"unit": "bytes per launch (average over the run; template instances of a kernel pooled)",
           "fetch_calibration": "raw (8 B/lane accesses: uncalibrated)", "code": sha, "csrc_sha1": csrc_digest(), "kernels": {}}
    for k in sorted(set(f) | set(w)):
        fl, fv = f.get(k, [0, 0.0])
        wl, wv = w.get(k, [0, 0.0])
        res["kernels"][k] = {"launches": int(max(fl, wl)), "fetch_bytes": 1024.0 * fv / max(fl, 1),
                             "write_bytes": 1024.0 * wv / max(wl, 1)}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in sorted(res["kernels"].items(), key=lambda kv: -(kv[1]["fetch_bytes"] + kv[1]["write_bytes"]) * kv[1]["launches"]):
        print("%-30s launches %6d  fetch %12.0f B  write %12.0f B" % (k, v["launches"], v["fetch_bytes"], v["write_bytes"]))


if __name__ == "__main__":
    main()
