#!/usr/bin/env python3
"""One line per bench.py JSON file: ms/step, solve, tree, leaf/small/big kernel time, roofline."""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
        continue
    s = d["symbolic"]
    print(f"{f}: {d['ms_per_step']:.3f} ms/step, solve {d['ms_per_linear_solve']:.3f} | {d['config']['ordering'][:12]} "
          f"relax {d['config']['amalgamation']['relax']}/{d['config']['amalgamation']['max_frontal_dim']} fronts {s['n_fronts']} "
          f"levels {s['n_levels']}/{s.get('n_upper_levels', '?')} big {s['n_big_fronts']} GF {s['factor_flops'] / 1e9:.2f} | leaf {d['factor_leaf_ms']:.3f} "
          f"small {d['factor_small_ms']:.3f} big {d['factor_big_ms']:.3f} backsolve {d['phases_ms']['ms_backsolve']:.3f} "
          f"asm {d['phases_ms']['ms_assemble_hessian']:.3f} | roof {d['roofline']['kernel']} {d['roofline']['frac']:.4f}")
    for k, v in d["kernels"].items():
        if v["ms_per_factorization"] > 0:
            print(f"      {k:40s} {v['ms_per_factorization']:.3f} ms  {v['launches_per_factorization']:.0f} launches")
