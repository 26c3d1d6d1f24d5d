#!/usr/bin/env python3
"""bench.py — LM iterations/sec and ms/linear-solve of the HIP hot path on MI355X.

A "step" is ONE Levenberg-Marquardt inner iteration at a fixed linearization point, with the
amortised linearize included: linearize -> Hessian panels -> damped multifrontal
Cholesky -> back-substitution -> 2 linear errors -> retract -> nonlinear error.  Nothing is
committed, so every step does identical work on data already resident in HBM.

  N = 1   workload = BAL Ladybug-1723 shape (1 723 cameras / 156 502 points / 678 718 observations,
          seeded synthetic: 70 % ring-local + 30 % half-lap revisit co-visibility); ordering: see `--ordering`
          (BASELINE config 3: "LM + METIS ordering" — `--ordering metis` feeds the reference's own METIS result).
          The line also carries `secondary`: pose3_100k ms/step on this one GPU — the N=1 anchor of the N>1 curve.
  N > 1   `python bench.py --gpus N` starts its own N ranks (torch.distributed.run) when it was not started
          under torchrun.  One process per GPU (torch.distributed, backend nccl = RCCL over xGMI): the ranks solve ONE
          100 000-pose Pose3 graph together ("100k-pose g2o @1/2/4/8", BASELINE's metric) — gsx_set_shard:
          subtrees of the Bayes tree dealt to the ranks, one all-reduce of the top ("cap") fronts per
          factorization; "scaling": "strong".  `--replicas` runs N independent seeded replicas of the N=1
          workload instead (weak scaling, no data-path collective).
          The sharded path has been rehearsed on ONE GPU only (several ranks sharing it, exchange over gloo):
          strong scaling on real xGMI and the RCCL branch of the exchange are unmeasured until the driver's
          multi-GPU run.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP64_PEAK_TFLOPS = 78.6    # MI355X FP64 vector = matrix peak (public spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=["bal1723", "bal49", "pose3_100k", "pose2_100k"],
                    help="default: bal1723 on one GPU (and for --replicas), pose3_100k for the sharded N > 1 run")
    ap.add_argument("--ordering", default=None, choices=[None, "schur", "schur_nd", "mindegree", "nd", "metis"],
                    help="the library's own orderings, or 'metis' = the order the REFERENCE's METIS_NodeND returns for this "
                         "seeded problem through Ordering::Metis (BASELINE config 3: 'LM + METIS ordering'), read from the "
                         "committed fixture tests/golden/metis_perm_<workload>_seed42.npy and handed over through "
                         "gsx_set_ordering exactly as GTSAM hands its params.ordering")
    ap.add_argument("--ordering-file", default=None, metavar="NPY",
                    help="elimination order from a file: int32 positions into the ascending-key variable table, or uint64 keys")
    ap.add_argument("--launch-check", action="store_true",
                    help="only the launch plumbing: rendezvous, barrier, MAX over ranks, rank 0's line — no GPU work "
                         "(tests/test_distributed_gloo.py runs `bench.py --gpus 2 --backend gloo --launch-check` on CPU)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="N=1: skip the pose-graph anchor (pose3_100k ms/step, the N=1 point of the sharded curve) that is "
                         "measured after the headline workload's timed region")
    ap.add_argument("--lam", type=float, default=1e-5)
    ap.add_argument("--hard-prior", action="store_true",
                    help="pose graphs: the prior on the first pose becomes noiseModel::Constrained::All (zero sigmas) — its "
                         "clique is eliminated with constraint pivots (constraint.hip); a measurement of that path, not the headline")
    ap.add_argument("--amalgamation", default=None, metavar="RELAX,MAXF",
                    help="relaxed clique amalgamation (gsx_set_amalgamation); default: the library's own choice "
                         "(GSX_AMALGAMATION_AUTO, what a drop-in caller gets), 0,128 = the reference's cliques")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="host threads of the CPU baseline (a one-GPU box's share of the host is 16 cores)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse the N>1 path on a box "
                         "with fewer GPUs than ranks)")
    ap.add_argument("--shard", action="store_true",
                    help="(the default for N>1) the ranks solve ONE problem together (gsx_set_shard: cap + subtrees, one "
                         "all-reduce of the cap per factorization); strong scaling")
    ap.add_argument("--replicas", action="store_true",
                    help="N>1: one independent seeded replica per rank instead (weak scaling, no data-path collective)")
    ap.add_argument("--shard-share", type=int, default=0, metavar="W",
                    help="N=1 only: time rank 0's share of a W-way sharded problem with the exchange stubbed out (rank 0 "
                         "carries the cap's damping, so its system stays positive definite): the compute on the critical "
                         "path of a W-GPU run, without the all-reduce")
    ap.add_argument("--shard-extra", default="none", metavar="WORKLOAD|none",
                    help="N>1 with --replicas: after the timed replicas, rank 0 also starts a separate N-rank sharded run "
                         "of this workload (own processes, hard time limit) and attaches its result under \"shard_run\"")
    ap.add_argument("--shard-extra-timeout", type=float, default=240.0)
    return ap.parse_args()


def make_problem(name, seed):
    from gtsam_petercdev_amd import datasets
    if name == "bal1723":
        return datasets.synth_bal_arrays(1723, 156502, 678718, seed=seed, long_range=0.3), "schur_nd"
    if name == "bal49":
        return datasets.synth_bal_arrays(49, 7776, 31843, seed=seed, long_range=0.3), "schur_nd"
    if name == "pose3_100k":
        return datasets.synth_manhattan_pose3(100000, seed=seed), "nd"
    return datasets.synth_manhattan_pose2(100000, seed=seed), "nd"


def load_ordering(arrays, workload, path=None):
    """Keys in elimination order from a fixture (int32 positions into the ascending-key variable table, or keys)."""
    if path is None:
        path = os.path.join(ROOT, "tests", "golden", f"metis_perm_{workload}_seed42.npy")
    a = np.load(path)
    if a.dtype == np.uint64:
        return a
    return arrays.var_keys[a.astype(np.int64)]


def self_launch(args):
    """`python bench.py --gpus N` with no torchrun environment: start the N ranks as a child job
    (python -m torch.distributed.run, one rank per GPU) BEFORE anything in this process touches the GPU, relay rank 0's
    JSON line, exit with the child's code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    line = None
    for l in proc.stdout:
        if l.startswith("{"):
            line = l.strip()
        else:
            sys.stderr.write(l)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    sys.exit(rc if rc != 0 or line is not None else 1)


def front_split(be, arrays, tile=32, mine=None):
    """Algorithmic bytes / flops of the factorization split by kernel class; the class of every front is
    the library's own (gsx_get_front_classes).  A front whose square is materialised moves 8 n^2 bytes; a LEAN leaf
    (childless, f <= 16, blocked parent) is the "fused landmark elimination" of §8(d): its Schur complement never exists, so
    it moves its H panel in and its L panel out (16 n f bytes) and the parent's gather reads that L panel once more.
    "small" = the LDS-class fronts incl. the medium ones (frontal panel in LDS, trailing block in HBM).
    mine (bool per front): a sharded rank's share — only the fronts it processes (own subtrees + cap) are counted."""
    parent, fronts = be.get_tree()
    cls_bits = be.front_classes()
    dims = arrays.var_dims
    nfr = len(fronts)
    F = np.array([float(dims[fv].sum()) for fv, _ in fronts])
    S1 = np.array([float(dims[sv].sum()) + 1.0 for _, sv in fronts])
    N = F + S1
    kind = cls_bits & 3          # 0 leaf kernel, 1 LDS, 2 blocked, 3 medium
    lean = (cls_bits & 8) != 0
    big = kind == 2
    out = dict(lpanel_bytes=0.0, big_syrk_flops=0.0, big_trsm_flops=0.0, big_potrf_flops=0.0, gather_bytes=0.0,
               n_lean=0, n_medium=0, n_tree=0)
    for k in ("leaf", "small", "big"):
        out.update({k + "_bytes": 0.0, k + "_flops": 0.0, "n_" + k: 0})
    for i in range(nfr):
        if mine is not None and not mine[i]:
            continue
        f, s1, n = F[i], S1[i], N[i]
        fl = f ** 3 / 3 + f * f * s1 + f * s1 * s1
        k = ("leaf", "small", "big", "small")[kind[i]]
        out[k + "_bytes"] += 16.0 * n * f if lean[i] else 8.0 * n * n
        out[k + "_flops"] += fl
        out["n_" + k] += 1
        out["n_lean"] += int(lean[i])
        out["n_medium"] += int(kind[i] == 3)
        out["n_tree"] += int((cls_bits[i] & 4) != 0)
        out["lpanel_bytes"] += 8.0 * f * n
        if k == "big":  # blocked path (csrc/bigfront.hip), rounds of <= 192 frontal columns: flops of each kernel
            nch = int(np.ceil(f / 192.0))
            chunk = min(192.0, np.ceil(np.ceil(f / nch) / tile) * tile)
            c0 = 0.0
            while c0 < f:
                w = min(chunk, f - c0)
                m = n - (c0 + w)
                out["big_potrf_flops"] += w ** 3 / 3                  # L11 = chol(A11) of the chunk (big_diag)
                out["big_trsm_flops"] += w * w * m                    # L21 = A21 L11^-T (big_rows)
                out["big_syrk_flops"] += w * m * (m + 1)              # A22 -= L21 L21' on the lower tile pairs (big_schur)
                c0 += w
            out["gather_bytes"] += 2 * 8.0 * n * (n + 1) / 2          # destination lower triangle read + written once
    for i, p in enumerate(parent):
        if mine is not None and not mine[i]:
            continue
        if p >= 0 and big[p]:  # what the big parent's gather reads of this child, once
            out["gather_bytes"] += 8.0 * N[i] * F[i] if lean[i] else 8.0 * S1[i] * (S1[i] + 1) / 2
    return out


def pmc_traffic(kernel, workload):
    """HBM bytes per launch (FETCH_SIZE + WRITE_SIZE) of `kernel` from the committed rocprofv3 --pmc passes of this
    same command (hardware counters cannot be read from inside the process; tools_pmc.sh + tools/pmc_summary.py made
    the file).  The newest round's file is used, and only while the sources it was captured from (`csrc_sha1`) are the
    sources of this tree: a stale file gives traffic = null, not an old number."""
    import glob
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from pmc_summary import csrc_digest
        now = csrc_digest(ROOT)
    except Exception:   # noqa: BLE001
        now = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}_pmc_traffic.json")), reverse=True):
        try:
            doc = json.load(open(path))
            # a pooled roofline entry ("front_small_kernel (+front_tree, front_medium)"): bytes per launch over all its kernels
            names = [kernel.split("(")[0].strip()]
            if "(+" in kernel:
                names += [n.strip() + "_kernel" for n in kernel.split("(+")[1].rstrip(")").split(",")]
            ks = [doc["kernels"][n] for n in names if n in doc["kernels"]]
            if not ks:
                raise KeyError(kernel)
        except (OSError, KeyError, ValueError):
            continue
        rel = os.path.relpath(path, ROOT)
        if now is None or doc.get("csrc_sha1") != now:
            return None, rel + " (stale: captured from other kernel sources)"
        launches = sum(k.get("launches", 1) for k in ks)
        return sum((k["fetch_bytes"] + k["write_bytes"]) * k.get("launches", 1) for k in ks) / max(launches, 1), rel
    return None, None


def run_shard_child(world, workload, backend, timeout_s):
    """The sharded solve as a child job of its own (torchrun, `world` ranks, its own rendezvous port), so that nothing it
    does — an error, a hang — can take the replica measurement down with it: killed as a process group at the limit."""
    import signal
    import subprocess
    port = int(os.environ.get("MASTER_PORT", "29500")) + 23
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--workload", workload, "--steps", "20", "--warmup", "3",
           "--no-cpu-baseline", "--backend", backend]
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "GROUP_RANK", "ROLE_RANK",
                        "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT",
                        "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    t0 = time.perf_counter()
    try:
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT,
                                start_new_session=True)
    except OSError as e:
        return {"error": f"could not start: {e!r}"}
    try:
        so, se = proc.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)   # exactly the group started above
        except OSError:
            pass
        so, se = proc.communicate()
        return {"error": f"no result within {timeout_s:.0f} s (killed)", "stderr_tail": (se or "")[-400:]}
    lines = [l for l in (so or "").splitlines() if l.startswith("{")]
    if proc.returncode != 0 or not lines:
        return {"error": f"exit code {proc.returncode}", "stderr_tail": (se or "")[-400:]}
    d = json.loads(lines[-1])
    return {"workload": workload, "n_gpus": d["n_gpus"], "scaling": d["scaling"], "value": d["value"], "unit": d["unit"],
            "ms_per_step": d["ms_per_step"], "ms_per_linear_solve": d["ms_per_linear_solve"], "shard": d.get("shard"),
            "phases_ms": d["phases_ms"], "wall_s": time.perf_counter() - t0}


def cpu_baseline(arrays, ordering, lam, workload, threads, iters):
    """The oracle ("port": a CPU restatement of the reference algorithm; GTSAM itself cannot be built here) on a bounded
    sample of the same workload, on this box's host cores — with the two loops the reference runs on TBB threaded (factors
    in linearize, independent subtrees in elimination / back-substitution), and single-threaded beside it.  A reported
    baseline, not the target.  The ONLY place bench.py touches oracle/."""
    from oracle import oracle as orc

    def cpu_run(nthreads, n):
        ob = orc.oracle_backend(arrays)
        ob.set_threads(nthreads)
        ob.set_ordering(ordering)
        ob.linearize()                       # (first touch of the allocator's arenas: not timed)
        ob.solve(lam, False, want_delta=False)
        ob.reset_timing()
        t_cpu = time.perf_counter()
        for _ in range(n):
            ob.linearize()
            ob.solve(lam, False, want_delta=False)
            ob.linear_error()
            ob.retract(None, commit=False)
        cpu_s = time.perf_counter() - t_cpu
        tm, _ = ob.timing()
        ob.close()
        return {"value": n / cpu_s, "ms_per_step": 1e3 * cpu_s / n,
                "ms_per_linear_solve": 1e3 * (tm["damp"] + tm["symbolic"] + tm["eliminate"] + tm["backsub"]) / n,
                "phases_s": {k: v / n for k, v in tm.items()}}
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    cores = max(1, min(threads, os.cpu_count() or 1))
    multi = cpu_run(cores, iters)
    single = cpu_run(1, max(1, iters - 1))
    return dict(
        multi, unit="LM iterations/s", cores=cores, kind="port",
        sample=f"{iters} identical LM inner iterations (linearize + damped multifrontal solve + 2 linear errors "
               f"+ retract + error) of the same {workload} problem and ordering, after one untimed iteration; "
               f"{cores} host threads on the loops the reference threads with TBB",
        cpu_model=cpu_model, host_cpus=os.cpu_count(), single_thread=dict(single, cores=1))


def anchor_ms_per_step(workload, ordering_name, device, lam, steps=20, warmup=3):
    """ms per LM inner iteration of `workload` on ONE GPU (a fresh handle, same step as the headline): the N=1 point of
    the sharded pose-graph curve, measured in the driver's own N=1 run."""
    from gtsam_petercdev_amd import _abi as A, _lib
    arrays, default_order = make_problem(workload, 42)
    be = _lib.product_backend(arrays, device=device)
    name = ordering_name or default_order
    if name == "metis":
        ordering = load_ordering(arrays, workload)
    else:
        ordering = be.compute_ordering({"schur": A.ORDER_SCHUR, "schur_nd": A.ORDER_SCHUR_ND,
                                        "mindegree": A.ORDER_MINDEGREE, "nd": A.ORDER_ND}[name])
    be.set_ordering(ordering)
    be.set_profiling(-1)
    for _ in range(warmup):
        be.lm_trial(True, lam, False)
    be.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        be.lm_trial(True, lam, False)
    be.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    st = be.stats()
    out = {"workload": workload, "ordering": name, "ms_per_step": ms, "lm_iterations_per_sec": 1e3 / ms, "steps": steps,
           "n_fronts": st["n_fronts"], "n_levels": st["n_levels"], "factor_flops": st["factor_flops"],
           "amalgamation": {"relax": st["amalgamation_relax"], "max_frontal_dim": int(st["amalgamation_max_frontal_dim"])}}
    be.close()
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)   # (does not return)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    from gtsam_petercdev_amd import _abi as A, _lib, distributed as D
    if args.launch_check:
        dist = D.init(args.backend, None)
        t0 = time.perf_counter()
        time.sleep(0.02 * (rank + 1))
        if dist is not None:
            dist.barrier()
        value, ms_step = D.aggregate_throughput(dist, args.steps, time.perf_counter() - t0)
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "gpus_arg": args.gpus, "value": value,
                              "ms_per_step": ms_step, "steps": args.steps}))
        if dist is not None:
            dist.destroy_process_group()
        return
    dist = None
    ndev = max(torch.cuda.device_count(), 1)
    device = local_rank % ndev  # == local_rank on a node with one GPU per rank
    if world > 1:
        torch.cuda.set_device(device)
        dist = D.init(args.backend, torch.device("cuda", device) if args.backend == "nccl" else None)
    red_dev = "cuda" if (dist is not None and args.backend == "nccl") else "cpu"

    sharded = (world > 1 and not args.replicas) or args.shard_share > 1
    if args.workload is None:
        args.workload = "pose3_100k" if (sharded and world > 1) else "bal1723"
    arrays, default_order = make_problem(args.workload, seed=42 if sharded else D.replica_seed(42))
    if args.hard_prior:
        from gtsam_petercdev_amd import _abi as _A
        pf = int(np.flatnonzero(arrays.f_type == _A.F_PRIOR)[0])
        assert (int(arrays.f_noise_kind[pf]) & _A.NOISE_BASE_MASK) == _A.NOISE_DIAGONAL, "the prior needs a diagonal model"
        arrays.noise[int(arrays.f_noise_ptr[pf]):int(arrays.f_noise_ptr[pf + 1])] = 0.0
    be = _lib.product_backend(arrays, device=device)
    exchange = {"calls": 0, "doubles": 0, "seconds": 0.0}
    exchange_path = None
    if sharded:
        if args.shard_share > 1:
            inner = lambda ptr, count: None
            shard_rank, shard_world = 0, args.shard_share
        else:
            # (a known-answer run of the in-place RCCL path first; all ranks fall back to a bounce buffer together if it fails)
            pp, pn = be.shard_probe_buffer()   # the known-answer run happens on memory the LIBRARY allocated
            inner, exchange_path = D.checked_allreduce(dist, torch.device("cuda", device), probe_ptr=pp, probe_count=pn)
            shard_rank, shard_world = rank, world

        def allreduce(ptr, count):
            t_ex = time.perf_counter()
            inner(ptr, count)
            exchange["seconds"] += time.perf_counter() - t_ex
            exchange["calls"] += 1
            exchange["doubles"] += count
        be.set_shard(shard_rank, shard_world, allreduce)
    ordering_name = args.ordering or default_order
    t0 = time.time()
    if args.ordering_file is not None or ordering_name == "metis":
        # an ordering computed by the caller and handed over through gsx_set_ordering — what GTSAM does with
        # params.ordering; the committed METIS fixtures are for the seed-42 problems
        ordering = load_ordering(arrays, args.workload, args.ordering_file)
        ordering_name = "file:" + os.path.basename(args.ordering_file) if args.ordering_file else \
            "metis (the reference's METIS_NodeND through Ordering::Metis, committed fixture)"
    else:
        okind = {"schur": A.ORDER_SCHUR, "schur_nd": A.ORDER_SCHUR_ND, "mindegree": A.ORDER_MINDEGREE,
                 "nd": A.ORDER_ND}[ordering_name]
        ordering = be.compute_ordering(okind)
    t_order = time.time() - t0
    if args.amalgamation is not None:   # (a new handle's default is the library's own choice)
        a, b = args.amalgamation.split(",")
        be.set_amalgamation(float(a), int(b))
    t0 = time.time()
    be.set_ordering(ordering)
    t_symbolic = time.time() - t0
    st0 = be.stats()
    relax, relax_maxf = st0["amalgamation_relax"], int(st0["amalgamation_max_frontal_dim"])
    lam = args.lam

    def step():
        # one LM inner iteration without the policy: linearize, H assembly, damped factorization, back-substitution,
        # both linearized errors, retract, nonlinear error of the trial point — one host synchronisation, as in
        # gsx_lm_optimize
        be.lm_trial(True, lam, False)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        be.synchronize()

    shard_check = None
    if sharded and world > 1:
        # the sharded trial against the plain single-GPU one on the same inputs (rank 0 runs the reference handle)
        e_sh = be.lm_trial(True, lam, False)
        if rank == 0:
            ref = _lib.product_backend(arrays, device=device)
            if args.amalgamation is not None:
                ref.set_amalgamation(relax, relax_maxf)
            ref.set_ordering(ordering)
            e_ref = ref.lm_trial(True, lam, False)
            ref.synchronize()
            t_ref = time.perf_counter()
            for _ in range(5):   # (the other ranks wait at their next collective meanwhile)
                ref.lm_trial(True, lam, False)
            ref.synchronize()
            single_gpu_ms = 1e3 * (time.perf_counter() - t_ref) / 5
            ref.close()
            shard_check = (max(abs(a - b) / max(abs(b), 1e-300) for a, b in zip(e_sh, e_ref)), single_gpu_ms)
    be.set_profiling(-1)   # no event timers inside the timed region (an event record costs ~5 us of stream time)
    for _ in range(args.warmup):
        step()
    be.reset_stats()
    barrier()
    for k in exchange:
        exchange[k] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    value, ms_step = D.aggregate_throughput(dist, args.steps, elapsed, device=red_dev)
    if sharded and world > 1:
        value /= world   # ONE job: its iterations per second, not the sum over replicas
    exchange_timed = dict(exchange)
    # phase times (HIP events around the phases) from a pass of their own, after the timed region
    be.set_profiling(0)
    be.reset_stats()
    for _ in range(min(args.steps, 10)):
        step()
    st = be.stats()
    ms_solve = (st["ms_factorize"] + st["ms_backsolve"]) / max(st["n_factorize"], 1)

    # ---- per-kernel timing pass (HIP events on the library's stream around every launch) ------------
    be.set_profiling(1)
    be.reset_stats()
    for _ in range(3):
        step()
    st_prof = be.stats()
    if sharded:   # this rank's share: its own subtrees + the cap (which every rank factors, in the blocked class)
        _, f_owner, _ = be.shard_info()
        split = front_split(be, arrays, mine=(f_owner == shard_rank) | (f_owner < 0))
    else:
        split = front_split(be, arrays)
    nfac = max(st_prof["n_factorize"], 1)
    phases = {k: st_prof[k] / nfac for k in ("ms_linearize", "ms_assemble_hessian", "ms_factorize", "ms_backsolve",
                                             "ms_linear_error", "ms_retract", "ms_error")}
    # every kernel of the path: (HIP-event name, rocprof kernel name, bound, algorithmic amount per
    # factorization — bytes for "hbm", flops for "mfma")
    kernels = [
        ("factor_leaf", "front_leaf_kernel", "hbm", split["leaf_bytes"]),
        ("factor_small", "front_small_kernel (+front_tree, front_medium)", "hbm", split["small_bytes"]),
        ("big_diag", "big_diag_kernel", "mfma", split["big_potrf_flops"]),
        ("big_rows", "big_rows_kernel", "mfma", split["big_trsm_flops"]),
        ("big_schur", "big_schur_kernel", "mfma", split["big_syrk_flops"]),
        ("big_gather", "big_gather_seg_kernel(+combine)", "hbm", split["gather_bytes"]),
        ("assemble_hessian", "assemble_h_kernel(+hessian_diag)", "hbm", st["jacobian_bytes"] + st["hessian_bytes"]),
        ("backsolve", "backsolve_kernel (all levels)", "hbm", split["lpanel_bytes"]),
    ]
    per_kernel = {}
    for ev, kname, bound, amount in kernels:
        avg_ms, n = be.kernel_time(ev)
        tot = avg_ms * n / nfac
        per_kernel[kname] = dict(ms_per_factorization=tot, launches_per_factorization=n / nfac, avg_launch_ms=avg_ms,
                                 bound=bound, algorithmic_per_factorization=amount)
    be.set_profiling(0)
    dom = max(per_kernel, key=lambda k: per_kernel[k]["ms_per_factorization"])
    dk = per_kernel[dom]
    per_launch = dk["algorithmic_per_factorization"] / max(dk["launches_per_factorization"], 1e-9)
    if dk["bound"] == "hbm":
        achieved = per_launch / (dk["avg_launch_ms"] * 1e-3) / 1e9 if dk["avg_launch_ms"] > 0 else 0.0
        peak, unit = HBM_PEAK_GBS, "GB/s"
    else:
        achieved = per_launch / (dk["avg_launch_ms"] * 1e-3) / 1e12 if dk["avg_launch_ms"] > 0 else 0.0
        peak, unit = FP64_PEAK_TFLOPS, "TFLOP/s"
    traffic, traffic_src = pmc_traffic(dom, args.workload)
    roofline = dict(kernel=dom, bound=dk["bound"], achieved=achieved, peak=peak, unit=unit, frac=achieved / peak,
                    traffic=traffic, traffic_source=traffic_src,
                    launches_per_factorization=dk["launches_per_factorization"],
                    avg_launch_ms=dk["avg_launch_ms"], algorithmic_per_launch=per_launch)
    tot_leaf = per_kernel["front_leaf_kernel"]["ms_per_factorization"]
    tot_small = per_kernel["front_small_kernel (+front_tree, front_medium)"]["ms_per_factorization"]
    tot_big = sum(per_kernel[k]["ms_per_factorization"] for k in per_kernel if k.startswith("big_"))

    out = {
        "metric": "lm_iterations_per_sec", "value": value, "unit": "LM iterations/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
        "scaling": "strong" if (sharded and world > 1) else "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": args.workload + ("+hard_prior" if args.hard_prior else ""), "shape": arrays.meta,
                   "ordering": ordering_name,
                   "amalgamation": {"relax": relax, "max_frontal_dim": relax_maxf,
                                    "chosen_by": "caller" if args.amalgamation is not None else "library"}, "lambda": lam,
                   "replicas": 1 if sharded else world,
                   "parallelism": (f"shard{world}" if world > 1 else f"rank 0 of shard{args.shard_share}, exchange "
                                   "stubbed") if sharded else f"replicas{world}"},
        "ms_per_linear_solve": ms_solve,
        "phases_ms": phases,
        "factor_leaf_ms": tot_leaf, "factor_small_ms": tot_small, "factor_big_ms": tot_big,
        "symbolic": {k: st[k] for k in ("n_fronts", "n_levels", "max_front_dim", "max_front_rows", "n_small_fronts",
                                        "n_big_fronts", "n_medium_fronts", "n_tree_fronts", "n_upper_levels", "factor_flops", "front_bytes", "lpanel_bytes",
                                        "jacobian_bytes", "hessian_bytes", "total_dim")},
        "front_split": split, "kernels": per_kernel, "host_ordering_s": t_order, "host_symbolic_s": t_symbolic,
        "roofline": roofline,
    }
    if sharded:
        info, owner, _ = be.shard_info()
        out["shard"] = dict(info, exchange_path=exchange_path, trial_vs_single_gpu_max_rel_diff=shard_check[0] if shard_check else None,
                            single_gpu_ms_per_step=shard_check[1] if shard_check else None, exchange_calls_per_step=exchange_timed["calls"] / args.steps,
                            exchange_mb_per_step=8e-6 * exchange_timed["doubles"] / args.steps,
                            exchange_ms_per_step=1e3 * exchange_timed["seconds"] / args.steps,
                            fronts_per_rank=[int((owner == r).sum()) for r in range(info["world"])])
        # roofline / kernels / front_split above are RANK 0's: its launches against the algorithmic amounts of the
        # fronts it processes (own subtrees + cap)
        out["roofline"]["of"] = f"rank {shard_rank} of {shard_world}: own subtree fronts + cap fronts"
    out["config"]["shape"] = {k: (v if not hasattr(v, "tolist") else None) for k, v in arrays.meta.items()
                              if not hasattr(v, "shape")}

    # ---- CPU baseline (rank 0; the other ranks of a sharded job wait at the barrier below) ---------
    if rank == 0 and not args.no_cpu_baseline and args.shard_share <= 1 and (world == 1 or sharded):
        out["cpu_baseline"] = cpu_baseline(arrays, ordering, lam, args.workload, args.cpu_threads, args.cpu_iters)
        out["speedup_vs_cpu_port_threads"] = value / out["cpu_baseline"]["value"]
        out["speedup_vs_cpu_port_1_thread"] = value / out["cpu_baseline"]["single_thread"]["value"]
    # ---- N=1: the pose-graph anchor (the N=1 point of the curve `--gpus N` measures on pose3_100k) ---------
    if rank == 0 and world == 1 and not sharded and not args.no_secondary and args.workload == "bal1723":
        be.close()
        try:
            out["secondary"] = [anchor_ms_per_step("pose3_100k", None, device, lam),
                                anchor_ms_per_step("pose3_100k", "metis", device, lam)]
        except Exception as e:   # noqa: BLE001 — the headline line must come out whatever happens here
            out["secondary"] = {"error": repr(e)}
    if dist is not None and sharded:
        dist.barrier()
    if dist is not None and not sharded and args.shard_extra != "none":
        # the other ranks wait on the host (a key of the rendezvous store), their GPUs idle, while rank 0's child job runs
        from torch.distributed.distributed_c10d import _get_default_store
        store = _get_default_store()
        if rank == 0:
            try:
                out["shard_run"] = run_shard_child(world, args.shard_extra, args.backend, args.shard_extra_timeout)
            except Exception as e:  # the headline line below must come out whatever happens here
                out["shard_run"] = {"error": repr(e)}
            store.set("gsx_shard_extra_done", "1")
        else:
            from datetime import timedelta
            store.wait(["gsx_shard_extra_done"], timedelta(seconds=args.shard_extra_timeout + 120))
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
