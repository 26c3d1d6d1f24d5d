"""Worker of the sharded-solve tests (tests/test_gpu_shard.py): torchrun starts WORLD_SIZE of these; every rank builds
the same seeded problem, makes its handle one shard of it (gsx_set_shard) and runs the solve collectively; beside it
every rank runs the plain single-GPU handle on the same inputs and compares.  The ranks share the box's one GPU, where
RCCL refuses duplicate devices, so the exchange goes over gloo (staged through host memory) — the same callback, the
same library path."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gtsam_petercdev_amd import _abi as A, _lib, datasets, distributed as D  # noqa: E402


def _hub_graph(nv, seed):
    """A Pose2 chain with random chords, two hubs with 22 / 40 neighbours and a dense cluster (leaf cliques with tall
    separators, blocked fronts): the irregular structures the seeded Manhattan worlds do not have."""
    import math
    from gtsam_petercdev_amd.graph import NonlinearFactorGraph, Values, Pose2, BetweenFactor, noiseModel
    rng = np.random.default_rng(seed)
    g, v = NonlinearFactorGraph(), Values()
    pos = np.cumsum(rng.normal(0.5, 0.2, (nv, 2)), axis=0)
    th = np.cumsum(rng.normal(0, 0.1, nv))
    for k in range(nv):
        v.insert(k, Pose2(pos[k, 0] + rng.normal(0, 0.03), pos[k, 1] + rng.normal(0, 0.03), th[k] + rng.normal(0, 0.01)))
    g.addPrior(0, Pose2(pos[0, 0], pos[0, 1], th[0]), noiseModel.Isotropic.Sigma(3, 0.1))
    pairs = {(k, k + 1) for k in range(nv - 1)}
    for a, b in rng.integers(0, nv, (nv, 2)):
        if a != b:
            pairs.add((int(min(a, b)), int(max(a, b))))
    for hub, deg in ((nv // 7, 22), (nv // 2, 40)):
        for b in rng.choice(nv, size=deg, replace=False):
            if int(b) != hub:
                pairs.add((min(hub, int(b)), max(hub, int(b))))
    c0 = 2 * nv // 3
    pairs |= {(a, b) for a in range(c0, c0 + 48) for b in range(a + 1, c0 + 48) if rng.random() < 0.6}
    sig = noiseModel.Diagonal.Sigmas(np.array([0.2, 0.2, 0.1]))
    for a, b in sorted(pairs):
        c, s_ = math.cos(th[a]), math.sin(th[a])
        dx, dy = pos[b, 0] - pos[a, 0], pos[b, 1] - pos[a, 1]
        g.add(BetweenFactor(a, b, Pose2(c * dx + s_ * dy + rng.normal(0, 0.05), -s_ * dx + c * dy + rng.normal(0, 0.05),
                                        th[b] - th[a] + rng.normal(0, 0.02)), sig))
    return g.to_arrays(v)


def problems():
    yield "pose2_hubs", _hub_graph(900, 8), A.ORDER_ND, 0.0
    yield "pose2", datasets.synth_manhattan_pose2(3000, seed=3), A.ORDER_ND, 0.0
    yield "pose3_relaxed", datasets.synth_manhattan_pose3(4000, seed=4), A.ORDER_ND, 0.5
    yield "bal", datasets.synth_bal_arrays(60, 4000, 18000, seed=5, long_range=0.3), A.ORDER_SCHUR_ND, 0.25
    yield "pose3_mindegree", datasets.synth_manhattan_pose3(1500, seed=6), A.ORDER_MINDEGREE, 0.0


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def main():
    rank, local_rank, world = D.env_rank()
    dist = D.init("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    allreduce = D.torch_allreduce(dist, dev)
    calls = {"n": 0, "doubles": 0}

    def counted(ptr, count):
        calls["n"] += 1
        calls["doubles"] += count
        allreduce(ptr, count)

    out = {}
    for name, arr, kind, relax in problems():
        ref = _lib.ProductBackend(arr, device=0)
        sh = _lib.ProductBackend(arr, device=0)
        ordering = ref.compute_ordering(kind)
        for be in (ref, sh):
            be.set_amalgamation(relax, 64)
        sh.set_shard(rank, world, counted)
        ref.set_ordering(ordering)
        sh.set_ordering(ordering)
        info, owner, owned = sh.shard_info()
        r = {"info": info, "owners": sorted(set(owner.tolist()))}
        # one damped step, both damping kinds
        r["error0"] = [sh.error(), ref.error()]
        ref.linearize()
        sh.linearize()
        r["hdiag"] = relerr(sh.hessian_diagonal(), ref.hessian_diagonal())
        steps = []
        for lam, diag in ((0.0, False), (1e-3, False), (1.0, True)):
            c0 = dict(calls)
            ds, dr = sh.solve(lam, diag), ref.solve(lam, diag)
            es, er = sh.linear_error(), ref.linear_error()
            ts, tr_ = sh.retract(None, commit=False), ref.retract(None, commit=False)
            steps.append({"delta": relerr(ds, dr), "lin": [list(es), list(er)], "trial": [ts, tr_],
                          "allreduces": calls["n"] - c0["n"]})
        r["steps"] = steps
        # whole LM runs: legacy (lambda I) and Ceres-style (diagonal damping) policies
        runs = []
        for params in (A.lm_params_legacy(), A.lm_params_ceres()):
            params.max_iterations = 8
            sh.set_values(arr.values)
            ref.set_values(arr.values)
            rs, rr = sh.lm_optimize(params), ref.lm_optimize(params)
            vs, vr = sh.get_values(), ref.get_values()
            runs.append({"accepted": [rs["trace_accepted"].tolist(), rr["trace_accepted"].tolist()],
                         "final": [rs["final_error"], rr["final_error"]], "initial": rs["initial_error"],
                         "trace": relerr(rs["trace_error"], rr["trace_error"]), "values": relerr(vs, vr)})
        r["lm"] = runs
        # Gauss-Newton goes through the same exchange points
        sh.set_values(arr.values)
        ref.set_values(arr.values)
        try:
            gs, gr = sh.gn_optimize(3), ref.gn_optimize(3)
            r["gn"] = [gs["final_error"], gr["final_error"]]
        except A.GsxError as e:  # (an indefinite GN step must at least fail on both alike)
            r["gn"] = str(e)
        # what a sharded handle does not offer fails loudly
        try:
            sh.marginal_covariance(int(arr.var_keys[0]))
            r["marginal_refused"] = False
        except A.GsxError as e:
            r["marginal_refused"] = e.status == A.GSX_E_STATE
        out[name] = r
        sh.close()
        ref.close()
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    if rank == 0:
        print(json.dumps({"world": world, "ranks": gathered, "allreduce_calls": calls}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
