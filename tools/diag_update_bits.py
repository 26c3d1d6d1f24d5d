"""Diagnosis: after gsx_update at config-5 size, which stage of partial-vs-full differs in bits?"""
import sys
import numpy as np
sys.path.insert(0, ".")
from gtsam_petercdev_amd import _abi as A, _lib, datasets

K, M, OBS = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (10000, 1000000, 4500000)
a1, id1 = datasets.synth_visual_slam(K, M, OBS, upto=K - 1)
a2, id2 = datasets.synth_visual_slam(K, M, OBS)
pos = {int(i): k for k, i in enumerate(id1)}
origin = np.array([pos.get(int(i), -1) for i in id2], dtype=np.int32)
so, old = a2.state_offsets(), set(int(k) for k in a1.var_keys)
new_idx = [i for i, k in enumerate(a2.var_keys) if int(k) not in old]
new_states = np.concatenate([a2.values[so[i]:so[i + 1]] for i in new_idx])


def partial_vs_full(pb, tag):
    so2 = a2.state_offsets()
    idx = np.nonzero(a2.var_types == A.VAR_POSE3)[0][-10:]
    cur = pb.get_values()
    rng = np.random.default_rng(5)
    states = np.concatenate([cur[so2[i]:so2[i + 1]] for i in idx])
    for k in range(len(idx)):
        states[12 * k + 9:12 * k + 12] += 1e-3 * rng.standard_normal(3)
    stp = pb.relinearize_partial(a2.var_keys[idx], states)
    jp = pb.jacobians()
    hp = pb.hessian_diagonal()
    dp = pb.solve(0.0, False)
    pb.linearize()
    jf = pb.jacobians()
    hf = pb.hessian_diagonal()
    df = pb.solve(0.0, False)
    print(tag, "fronts redone", stp["n_fronts_reeliminated"], "of", stp["n_fronts"], "| jac equal", np.array_equal(jp, jf),
          "n diff", int(np.sum(jp != jf)), "| hdiag equal", np.array_equal(hp, hf), "| step equal", np.array_equal(dp, df),
          "rel", float(np.linalg.norm(dp - df) / np.linalg.norm(df)), flush=True)


# (1) a fresh handle on the grown graph
import os
if os.environ.get("DIAG_SKIP_FRESH") is None:
    fb = _lib.product_backend(a2)
    fb.set_ordering(fb.compute_ordering(A.ORDER_SCHUR_ND))
    fb.linearize()
    fb.solve(0.0, False, want_delta=False)
    partial_vs_full(fb, "fresh  ")
    fb.close()
# (2) the updated handle
pb = _lib.product_backend(a1)
pb.set_ordering(pb.compute_ordering(A.ORDER_SCHUR_ND))
pb.linearize()
pb.solve(0.0, False, want_delta=False)
pb.update(a2, origin, new_states)
print({k: v for k, v in pb.stats().items() if k.startswith("n_")}, flush=True)
j1 = pb.jacobians()
pb.linearize()
j2 = pb.jacobians()
print("kept blocks after update equal to a re-linearization:", np.array_equal(j1, j2), int(np.sum(j1 != j2)), flush=True)
pb.solve(0.0, False, want_delta=False)
partial_vs_full(pb, "updated")

# ---- (3) locate the first front whose conditional differs between the partial and the full path ----
if os.environ.get("DIAG_LOCATE") is not None:
    parent, fronts = pb.get_tree()
    front_of = np.zeros(a2.n_vars, np.int64)
    for c, (fv, _) in enumerate(fronts):
        for v in fv:
            front_of[v] = c
    so2 = a2.state_offsets()
    idx = np.nonzero(a2.var_types == A.VAR_POSE3)[0][-10:]
    cur = pb.get_values()
    rng = np.random.default_rng(7)
    states = np.concatenate([cur[so2[i]:so2[i + 1]] for i in idx])
    for k in range(len(idx)):
        states[12 * k + 9:12 * k + 12] += 1e-3 * rng.standard_normal(3)
    # dirty fronts (host): factors touching a moved variable -> their variables -> fronts -> ancestors
    moved = set(int(i) for i in idx)
    fk = a2.f_key_ptr
    dvar = set()
    for f in range(a2.n_factors):
        vs = a2.f_vars[fk[f]:fk[f + 1]]
        if any(int(v) in moved for v in vs):
            dvar.update(int(v) for v in vs)
    dirty = set()
    for v in dvar:
        c = int(front_of[v])
        while c >= 0 and c not in dirty:
            dirty.add(c)
            c = parent[c]
    dirty = sorted(dirty)
    print("dirty fronts (host)", len(dirty), flush=True)
    pb.relinearize_partial(a2.var_keys[idx], states)
    cp = {c: pb.conditional(c) for c in dirty}
    pb.linearize()
    pb.solve(0.0, False, want_delta=False)
    classes = pb.front_classes()
    nbad = 0
    for c in dirty:          # ascending id = children first
        cf = pb.conditional(c)
        if not np.array_equal(cp[c], cf):
            kids = [k for k in range(len(parent)) if False]
            print("front", c, "class", int(classes[c]), "frontal vars", len(fronts[c][0]), "sep vars", len(fronts[c][1]),
                  "shape", cf.shape, "max diff", float(np.max(np.abs(cp[c] - cf))), "parent", parent[c],
                  "parent dirty", parent[c] in set(dirty), flush=True)
            nbad += 1
            if nbad >= 6:
                break
    print("differing fronts shown", nbad, flush=True)
