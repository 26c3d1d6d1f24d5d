"""The multi-process path of bench.py (one process per GPU, barrier, MAX over ranks, rank-0 JSON line)
exercised with world_size 2 on the gloo backend — no GPU needed."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_replicas_gloo():
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_gloo_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["world"] == 2 and len(r["ranks"]) == 2
    (r0, n0, s0, f0, e0), (r1, n1, s1, f1, e1) = sorted(r["ranks"])
    assert (r0, r1) == (0, 1)
    assert n0 == n1 and s0 != s1            # same shape, different seeded replica per rank
    slowest = max(e0, e1)
    assert abs(r["ms_per_step"] - 1e3 * slowest / 4) < 1e-6 * max(1.0, r["ms_per_step"])
    assert abs(r["value"] - 2 * 4 / slowest) < 1e-6 * r["value"]   # whole-job rate = world * steps / max time


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment starts the two ranks itself (before anything touches a
    GPU), relays rank 0's line and exits with the child's code — the launch plumbing only (`--launch-check`, gloo)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                          "--launch-check", "--steps", "4"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["launch_check"] and r["n_gpus"] == 2 and r["gpus_arg"] == 2
    assert r["ms_per_step"] >= 1e3 * 0.04 / 4 - 1e-6        # the slower rank (0.04 s) sets the time
    # and under an explicit torchrun (the driver's way) bench.py must NOT launch again
    port = 31500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--launch-check"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
