"""Build csrc/libgsx.so with hipcc for gfx950 (in-tree, so the .so travels to the GPU box)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libgsx.so")
SOURCES = ["problem.cpp", "ordering.cpp", "nd.cpp", "symbolic.cpp", "io.cpp", "lm_policy.cpp", "kernels.hip", "bigfront.hip", "constraint.hip", "solver.hip"]
HEADERS = ["gsx_internal.h", "kernels.h", "device_geometry.h", os.path.join("..", "..", "include", "gsx.h")]


def _stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in SOURCES + HEADERS)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    # GSX_BUILD_DIR: an experiment build (GSX_STAMP / GSX_EXTRA_DEFINES) into a directory of its own, e.g. build/exp_stamp,
    # leaving the product library alone (tools_exp.sh copies it over on the GPU box)
    out_dir = os.environ.get("GSX_BUILD_DIR")
    out = os.path.join(out_dir, "libgsx.so") if out_dir else OUT
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    elif not force and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for s in SOURCES:
        o = os.path.join(out_dir or CSRC, os.path.splitext(s)[0] + ".o")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
               "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value", "-c", os.path.join(CSRC, s), "-o", o]
        if os.environ.get("GSX_STAMP") and s.endswith(".hip"):
            cmd.insert(1, "-DGSX_STAMP")
        if os.environ.get("GSX_EXTRA_DEFINES"):   # (experiment builds: tools_exp.sh)
            for d in os.environ["GSX_EXTRA_DEFINES"].split():
                cmd.insert(1, "-D" + d)
        if s.endswith(".cpp"):
            cmd.insert(1, "-x")
            cmd.insert(2, "c++")
            cmd.remove("--offload-arch=gfx950")
            cmd.remove("-munsafe-fp-atomics")
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        objs.append(o)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
