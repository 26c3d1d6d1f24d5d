"""On-disk formats (BAL, g2o) and seeded synthetic generators of the BASELINE shapes.

Host-side data preparation only (numpy): everything here produces the flat
`ProblemArrays` of include/gsx.h; no number on the hot path is computed here.

Formats follow the reference's readers:
  BAL   gtsam/sfm/SfmData.cpp:79-97,189-245  (values parsed as float, then widened;
        measurement stored as (u, -v); pose converted by openGL2gtsam)
  g2o   gtsam/slam/dataset.cpp:216-296,505-633 (2-D), :756-863 (3-D)
The real Ladybug / w10000 files are not in the reference tree and there is no
network, so the named shapes are generated (seeded) instead.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np

from . import _abi as A
from .graph import P as _P, Rot3

# ------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------


def _expmap_so3(w: np.ndarray) -> np.ndarray:
    """Batched SO3::Expmap (gtsam/geometry/SO3.cpp:61-96); w: (n,3) -> (n,3,3)."""
    w = np.atleast_2d(np.asarray(w, dtype=float))
    th2 = np.einsum("ij,ij->i", w, w)
    th = np.sqrt(th2)
    small = th2 <= np.finfo(float).eps
    ths = np.where(small, 1.0, th)
    a = np.where(small, 1.0 - th2 / 6.0, np.sin(ths) / ths)
    s2 = np.sin(ths / 2.0)
    b = np.where(small, 0.5 - th2 / 24.0, 2.0 * s2 * s2 / np.where(small, 1.0, th2))
    W = np.zeros((w.shape[0], 3, 3))
    W[:, 0, 1], W[:, 0, 2] = -w[:, 2], w[:, 1]
    W[:, 1, 0], W[:, 1, 2] = w[:, 2], -w[:, 0]
    W[:, 2, 0], W[:, 2, 1] = -w[:, 1], w[:, 0]
    return np.eye(3)[None] + a[:, None, None] * W + b[:, None, None] * (W @ W)


def _pose3_expmap(xi: np.ndarray):
    """Batched Pose3::Expmap (gtsam/geometry/Pose3.cpp:184-222) -> (R (n,3,3), t (n,3))."""
    xi = np.atleast_2d(xi)
    w, v = xi[:, :3], xi[:, 3:]
    th2 = np.einsum("ij,ij->i", w, w)
    near = th2 <= 1e-5
    th = np.sqrt(np.where(near, 1.0, th2))
    a = np.where(near, 1.0 - th2 / 6.0, np.sin(th) / th)
    s2 = np.sin(th / 2.0)
    b = np.where(near, 0.5 - th2 / 24.0, 2.0 * s2 * s2 / np.where(near, 1.0, th2))
    c = np.where(near, 1.0 / 6.0 - th2 / 120.0, (1.0 - a) / np.where(near, 1.0, th2))
    W = np.zeros((w.shape[0], 3, 3))
    W[:, 0, 1], W[:, 0, 2] = -w[:, 2], w[:, 1]
    W[:, 1, 0], W[:, 1, 2] = w[:, 2], -w[:, 0]
    W[:, 2, 0], W[:, 2, 1] = -w[:, 1], w[:, 0]
    R = np.eye(3)[None] + a[:, None, None] * W + b[:, None, None] * (W @ W)
    wv = np.cross(w, v)
    wwv = np.cross(w, wv)
    t = v + b[:, None] * wv + c[:, None] * wwv
    return R, t


def _csr(counts):
    return np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)


# ------------------------------------------------------------------------------------------
# BAL
# ------------------------------------------------------------------------------------------
class SfmData:
    """cameras: (nc,17) states [R9 t3 f k1 k2 u0 v0]; points (np,3); observations sorted by track
    (point) in file order within a track: cam_idx, pt_idx, uv (GTSAM convention: (u, -v_file))."""

    def __init__(self, cameras, points, cam_idx, pt_idx, uv):
        self.cameras, self.points = cameras, points
        self.cam_idx, self.pt_idx, self.uv = cam_idx, pt_idx, uv

    def numberCameras(self):
        return self.cameras.shape[0]

    def numberTracks(self):
        return self.points.shape[0]


def read_bal(path: str) -> SfmData:
    """SfmData::FromBalFile (gtsam/sfm/SfmData.cpp:189-245)."""
    tok = open(path).read().split()
    nc, npts, nobs = int(tok[0]), int(tok[1]), int(tok[2])
    o = 3
    obs = np.array(tok[o:o + 4 * nobs], dtype=np.float64).reshape(nobs, 4)
    ci = obs[:, 0].astype(np.int64)
    pj = obs[:, 1].astype(np.int64)
    uvf = obs[:, 2:4].astype(np.float32).astype(np.float64)  # "float u, v;"
    o += 4 * nobs
    cam = np.array(tok[o:o + 9 * nc], dtype=np.float64).reshape(nc, 9).astype(np.float32).astype(np.float64)
    o += 9 * nc
    pts = np.array(tok[o:o + 3 * npts], dtype=np.float64).reshape(npts, 3).astype(np.float32).astype(np.float64)
    # openGL2gtsam (SfmData.cpp:79-85): wRc = R' * diag(1,-1,-1); t_w = R' * (-t)
    R = _expmap_so3(cam[:, :3])
    R90 = np.diag([1.0, -1.0, -1.0])
    wRc = np.transpose(R, (0, 2, 1)) @ R90
    tw = np.einsum("nji,nj->ni", R, -cam[:, 3:6])
    cams = np.concatenate([wRc.reshape(nc, 9), tw, cam[:, 6:9], np.zeros((nc, 2))], axis=1)
    # tracks[j].measurements in file order: stable sort by point
    order = np.argsort(pj, kind="stable")
    uv = np.stack([uvf[:, 0], -uvf[:, 1]], axis=1)
    return SfmData(cams, pts, ci[order], pj[order], uv[order])


def bal_arrays(sfm: SfmData, camera_keys=None, point_keys=None, priors: bool = False,
               sigma: Optional[float] = None) -> A.ProblemArrays:
    """The graph of tests/testGeneralSFMFactorB.cpp:44-63 / examples/SFMExample_bal.cpp:55-68:
    one GeneralSFMFactor per observation in track order (unit noise unless sigma), optional
    priors Isotropic(9,0.1) on camera 0 and Isotropic(3,0.1) on point 0 appended last."""
    nc, npts = sfm.numberCameras(), sfm.numberTracks()
    nobs = sfm.cam_idx.size
    ck = np.arange(nc, dtype=np.uint64) if camera_keys is None else np.asarray(camera_keys, dtype=np.uint64)
    pk = (np.uint64(ord("p")) << np.uint64(56)) + np.arange(npts, dtype=np.uint64) if point_keys is None \
        else np.asarray(point_keys, dtype=np.uint64)
    keys = np.concatenate([ck, pk])
    order = np.argsort(keys, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    types = np.concatenate([np.full(nc, A.VAR_CAMERA), np.full(npts, A.VAR_VECTOR)])[order]
    dims = np.concatenate([np.full(nc, 9), np.full(npts, 3)])[order]
    cam_var, pt_var = rank[:nc], rank[nc:]
    nprior = 2 if priors else 0
    nf = nobs + nprior
    f_type = np.full(nf, A.F_SFM, np.int32)
    f_rows = np.full(nf, 2, np.int32)
    nkeys = np.full(nf, 2, np.int64)
    fv = np.stack([cam_var[sfm.cam_idx], pt_var[sfm.pt_idx]], axis=1).reshape(-1)
    meas = [sfm.uv.reshape(-1)]
    meas_n = np.full(nf, 2, np.int64)
    nkind = np.full(nf, A.NOISE_UNIT if sigma is None else A.NOISE_ISOTROPIC, np.int32)
    noise_n = np.full(nf, 0 if sigma is None else 1, np.int64)
    noise = [np.full(nobs if sigma is not None else 0, sigma if sigma is not None else 0.0)]
    if priors:
        f_type[nobs:] = A.F_PRIOR
        f_rows[nobs:] = [9, 3]
        nkeys[nobs:] = 1
        fv = np.concatenate([fv, [cam_var[0], pt_var[0]]])
        meas += [sfm.cameras[0], sfm.points[0]]
        meas_n[nobs:] = [17, 3]
        nkind[nobs:] = A.NOISE_ISOTROPIC
        noise_n[nobs:] = 1
        noise += [np.array([0.1, 0.1])]
    # packed values in ascending key order
    states = [None] * (nc + npts)
    vals = np.concatenate([sfm.cameras.reshape(-1), sfm.points.reshape(-1)])
    if not np.array_equal(order, np.arange(order.size)):
        sd = np.concatenate([np.full(nc, 17), np.full(npts, 3)])
        off = _csr(sd)
        vals = np.concatenate([vals[off[i]:off[i + 1]] for i in order])
    return A.ProblemArrays(
        var_keys=keys[order], var_types=types, var_dims=dims, f_type=f_type, f_rows=f_rows,
        f_key_ptr=_csr(nkeys), f_vars=fv, f_meas_ptr=_csr(meas_n), meas=np.concatenate(meas),
        f_noise_kind=nkind, f_noise_ptr=_csr(noise_n), noise=np.concatenate(noise), values=vals,
        meta=dict(kind="bal", n_cams=nc, n_points=npts, n_obs=nobs, cam_vars=cam_var, pt_vars=pt_var))


def synth_bal(n_cams: int, n_points: int, n_obs: int, seed: int = 42, long_range: float = 0.0,
              pixel_sigma: float = 0.5, priors: bool = True, camera_symbols: bool = False):
    """Seeded BAL-shaped problem: cameras on a ring r=30 looking at the origin,
    Cal3Bundler(f~800+-100, k1~-1e-7, k2~1e-13), points U[-8,8]^3, each point seen by k>=2 cameras
    (sum k = n_obs exactly): a ring-local window of consecutive cameras, a `long_range` fraction of
    the observations coming from the cameras half a lap away instead (loop-closure revisits).  Returns (truth SfmData,
    perturbed-initial SfmData)."""
    rng = np.random.default_rng(seed)
    phi = 2 * np.pi * np.arange(n_cams) / n_cams
    pos = np.stack([30 * np.cos(phi), 30 * np.sin(phi), rng.uniform(-2, 2, n_cams)], axis=1)
    zc = -pos / np.linalg.norm(pos, axis=1, keepdims=True)  # optical axis -> origin
    up = np.array([0.0, 0.0, 1.0])
    xc = np.cross(np.broadcast_to(up, zc.shape), zc)
    xc /= np.linalg.norm(xc, axis=1, keepdims=True)
    yc = np.cross(zc, xc)
    R = np.stack([xc, yc, zc], axis=2)  # columns = camera axes in world
    f = 800 + rng.uniform(-100, 100, n_cams)
    k1 = -1e-7 * rng.uniform(0.5, 1.5, n_cams)
    k2 = 1e-13 * rng.uniform(0.5, 1.5, n_cams)
    cams = np.concatenate([R.reshape(n_cams, 9), pos, f[:, None], k1[:, None], k2[:, None],
                           np.zeros((n_cams, 2))], axis=1)
    pts = rng.uniform(-8, 8, (n_points, 3))
    # observation counts: 2 each + the rest dealt at random, capped at n_cams
    k = np.full(n_points, 2, np.int64)
    extra = n_obs - 2 * n_points
    if extra < 0:
        raise ValueError("need n_obs >= 2 n_points")
    cap = min(n_cams, 64)
    while extra > 0:
        add = rng.multinomial(extra, np.full(n_points, 1.0 / n_points))
        newk = np.minimum(k + add, cap)
        extra -= int((newk - k).sum())
        k = newk
        if np.all(k == cap):
            raise ValueError("too many observations for this many cameras")
    pt_idx = np.repeat(np.arange(n_points), k)
    start = rng.integers(0, n_cams, n_points)
    within = np.arange(pt_idx.size) - np.repeat(_csr(k)[:-1], k)
    cam_idx = (np.repeat(start, k) + within) % n_cams
    if long_range > 0:
        far = rng.random(pt_idx.size) < long_range
        far &= within >= 1  # keep the first observation local
        # "revisit" structure (a vehicle passing the same place twice, as in the Ladybug sequences):
        # a far observation comes from the camera half a lap away from the local window, so the
        # reduced camera graph is a band plus a shifted band, not a dense blob
        cam_idx = np.where(far, (cam_idx + n_cams // 2) % n_cams, cam_idx)
        # remove duplicate (cam, point) pairs by re-drawing deterministically
        for _ in range(8):
            key = pt_idx * n_cams + cam_idx
            _, first = np.unique(key, return_index=True)
            dup = np.ones(key.size, bool)
            dup[first] = False
            if not dup.any():
                break
            cam_idx = np.where(dup, (cam_idx + 1 + rng.integers(0, n_cams - 1, key.size)) % n_cams, cam_idx)
        o = np.lexsort((cam_idx, pt_idx))
        cam_idx, pt_idx = cam_idx[o], pt_idx[o]
    # project (A.1 of SURVEY): q = R'(p - t)
    Rc = R[cam_idx]
    q = np.einsum("nji,nj->ni", Rc, pts[pt_idx] - pos[cam_idx])
    assert np.all(q[:, 2] > 0)
    u, v = q[:, 0] / q[:, 2], q[:, 1] / q[:, 2]
    r = u * u + v * v
    g = 1 + (k1[cam_idx] + k2[cam_idx] * r) * r
    uv = np.stack([f[cam_idx] * g * u, f[cam_idx] * g * v], axis=1) + rng.normal(0, pixel_sigma, (cam_idx.size, 2))
    truth = SfmData(cams, pts, cam_idx, pt_idx, uv)
    # perturbed initial estimate: pose tangent N(0,0.01), f N(0,1), points N(0,0.05)
    xi = rng.normal(0, 0.01, (n_cams, 6))
    dR, dt = _pose3_expmap(xi)
    R0 = R @ dR
    t0 = pos + np.einsum("nij,nj->ni", R, dt)
    cams0 = cams.copy()
    cams0[:, :9] = R0.reshape(n_cams, 9)
    cams0[:, 9:12] = t0
    cams0[:, 12] += rng.normal(0, 1.0, n_cams)
    pts0 = pts + rng.normal(0, 0.05, pts.shape)
    init = SfmData(cams0, pts0, cam_idx, pt_idx, uv)
    return truth, init


def synth_bal_arrays(n_cams, n_points, n_obs, seed=42, long_range=0.0, priors=True) -> A.ProblemArrays:
    _, init = synth_bal(n_cams, n_points, n_obs, seed, long_range)
    arr = bal_arrays(init, priors=priors)
    arr.meta.update(seed=seed, long_range=long_range)
    return arr


# ------------------------------------------------------------------------------------------
# pose graphs
# ------------------------------------------------------------------------------------------
def _pose_graph_arrays(var_type, states, edges_i, edges_j, meas, sigmas, prior_sigmas, keys=None,
                       noise_R=None) -> A.ProblemArrays:
    """Between factors (Diagonal sigmas or per-edge Gaussian sqrt-information) + a prior on pose 0
    appended last, as examples/Pose2SLAMExample_g2o.cpp:65-67 / Pose3SLAMExample_g2o.cpp:42-48 do."""
    n = states.shape[0]
    d = 3 if var_type == A.VAR_POSE2 else 6
    m = edges_i.size
    keys = np.arange(n, dtype=np.uint64) if keys is None else np.asarray(keys, np.uint64)
    nf = m + 1
    f_type = np.full(nf, A.F_BETWEEN, np.int32)
    f_type[m] = A.F_PRIOR
    nkeys = np.full(nf, 2, np.int64)
    nkeys[m] = 1
    fv = np.concatenate([np.stack([edges_i, edges_j], axis=1).reshape(-1), [0]])
    sd = states.shape[1]
    meas_all = np.concatenate([meas.reshape(-1), states[0]])
    if noise_R is None:
        nkind = np.full(nf, A.NOISE_DIAGONAL, np.int32)
        noise_n = np.full(nf, d, np.int64)
        noise = np.concatenate([np.broadcast_to(sigmas, (m, d)).reshape(-1), prior_sigmas])
    else:
        nkind = np.full(nf, A.NOISE_GAUSSIAN, np.int32)
        nkind[m] = A.NOISE_DIAGONAL
        noise_n = np.full(nf, d * d, np.int64)
        noise_n[m] = d
        noise = np.concatenate([noise_R.reshape(-1), prior_sigmas])
    return A.ProblemArrays(
        var_keys=keys, var_types=np.full(n, var_type), var_dims=np.full(n, d), f_type=f_type,
        f_rows=np.full(nf, d, np.int32), f_key_ptr=_csr(nkeys), f_vars=fv,
        f_meas_ptr=_csr(np.full(nf, sd, np.int64)), meas=meas_all, f_noise_kind=nkind,
        f_noise_ptr=_csr(noise_n), noise=noise, values=states.reshape(-1),
        meta=dict(kind="pose2" if d == 3 else "pose3", n_poses=n, n_edges=m))


def _manhattan_walk(n_poses, rng, p_turn=0.3):
    """Grid random walk: returns integer positions (n,2), headings (n,) in quarter turns, and
    loop-closure candidate pairs (i<j) of poses sharing a grid cell."""
    pos = np.zeros((n_poses, 2), np.int64)
    head = np.zeros(n_poses, np.int64)
    dirs = np.array([[1, 0], [0, 1], [-1, 0], [0, -1]])
    turns = rng.random(n_poses) < p_turn
    sign = rng.integers(0, 2, n_poses) * 2 - 1
    cells = {}
    pairs = []
    for i in range(1, n_poses):
        h = head[i - 1]
        if turns[i]:
            h = (h + sign[i]) % 4
        head[i] = h
        pos[i] = pos[i - 1] + dirs[h]
        c = (int(pos[i, 0]), int(pos[i, 1]))
        lst = cells.get(c)
        if lst is None:
            cells[c] = [i]
        else:
            if i - lst[-1] > 10:
                pairs.append((lst[-1], i))
            lst.append(i)
    return pos, head, np.array(pairs, np.int64).reshape(-1, 2)


def synth_manhattan_pose2(n_poses=10000, seed=7, closure_prob=0.6, init_sigma=0.05) -> A.ProblemArrays:
    """Manhattan-world Pose2 graph: odometry chain + loop closures between poses
    that revisit a grid cell; EDGE_SE2 noise sigma=(0.05,0.05,0.02); initial = truth (+) N(0,init_sigma)
    per tangent dim; prior Diagonal::Variances(1e-6,1e-6,1e-8) on pose 0."""
    rng = np.random.default_rng(seed)
    pos, head, pairs = _manhattan_walk(n_poses, rng)
    truth = np.stack([pos[:, 0].astype(float), pos[:, 1].astype(float), head * (np.pi / 2)], axis=1)
    truth[:, 2] = np.arctan2(np.sin(truth[:, 2]), np.cos(truth[:, 2]))
    keep = rng.random(pairs.shape[0]) < closure_prob
    pairs = pairs[keep]
    ei = np.concatenate([np.arange(n_poses - 1), pairs[:, 0]])
    ej = np.concatenate([np.arange(1, n_poses), pairs[:, 1]])
    sig = np.array([0.05, 0.05, 0.02])

    def between(a, b):
        c, s = np.cos(a[:, 2]), np.sin(a[:, 2])
        dx, dy = b[:, 0] - a[:, 0], b[:, 1] - a[:, 1]
        th = b[:, 2] - a[:, 2]
        return np.stack([c * dx + s * dy, -s * dx + c * dy, np.arctan2(np.sin(th), np.cos(th))], axis=1)

    z = between(truth[ei], truth[ej]) + rng.normal(0, 1, (ei.size, 3)) * sig
    init = truth.copy()
    d = rng.normal(0, init_sigma, (n_poses, 3))
    c, s = np.cos(truth[:, 2]), np.sin(truth[:, 2])
    init[:, 0] += c * d[:, 0] - s * d[:, 1]
    init[:, 1] += s * d[:, 0] + c * d[:, 1]
    init[:, 2] += d[:, 2]
    init[0] = truth[0]
    arr = _pose_graph_arrays(A.VAR_POSE2, init, ei, ej, z, sig, np.sqrt([1e-6, 1e-6, 1e-8]))
    arr.meta.update(seed=seed)
    return arr


def synth_manhattan_pose3(n_poses=100000, seed=7, closure_prob=0.45, init_sigma=0.02) -> A.ProblemArrays:
    """Planar-SE(3) Manhattan graph: the Pose2 walk lifted to Pose3, noise
    sigma = (0.02 rad x3, 0.05 m x3), prior Diagonal::Variances(1e-6 x3, 1e-4 x3) on pose 0."""
    rng = np.random.default_rng(seed)
    pos, head, pairs = _manhattan_walk(n_poses, rng)
    yaw = head * (np.pi / 2)
    Rw = _expmap_so3(np.stack([np.zeros(n_poses), np.zeros(n_poses), yaw], axis=1))
    tw = np.stack([pos[:, 0].astype(float), pos[:, 1].astype(float), np.zeros(n_poses)], axis=1)
    keep = rng.random(pairs.shape[0]) < closure_prob
    pairs = pairs[keep]
    ei = np.concatenate([np.arange(n_poses - 1), pairs[:, 0]])
    ej = np.concatenate([np.arange(1, n_poses), pairs[:, 1]])
    sig = np.array([0.02, 0.02, 0.02, 0.05, 0.05, 0.05])
    Ri, Rj = Rw[ei], Rw[ej]
    Rz = np.transpose(Ri, (0, 2, 1)) @ Rj
    tz = np.einsum("nji,nj->ni", Ri, tw[ej] - tw[ei])
    dR, dt = _pose3_expmap(rng.normal(0, 1, (ei.size, 6)) * sig)
    Rz, tz = Rz @ dR, tz + np.einsum("nij,nj->ni", Rz, dt)
    meas = np.concatenate([Rz.reshape(-1, 9), tz], axis=1)
    dR0, dt0 = _pose3_expmap(rng.normal(0, init_sigma, (n_poses, 6)))
    dR0[0], dt0[0] = np.eye(3), 0
    R0 = Rw @ dR0
    t0 = tw + np.einsum("nij,nj->ni", Rw, dt0)
    init = np.concatenate([R0.reshape(-1, 9), t0], axis=1)
    arr = _pose_graph_arrays(A.VAR_POSE3, init, ei, ej, meas, sig, np.sqrt([1e-6] * 3 + [1e-4] * 3))
    arr.meta.update(seed=seed)
    return arr


# ------------------------------------------------------------------------------------------
# g2o (gtsam/slam/dataset.cpp)
# ------------------------------------------------------------------------------------------
def read_g2o(path: str, is3D: bool = False, add_prior: bool = True) -> A.ProblemArrays:
    """readG2o: 2-D VERTEX_SE2/EDGE_SE2 (dataset.cpp:216-296,505-633; info order I11 I12 I13 I22 I23 I33);
    3-D VERTEX_SE3:QUAT / EDGE_SE3:QUAT (dataset.cpp:756-863; information given in (t,R) order and
    permuted to GTSAM's (R,t), :850-856).  Noise = Gaussian::Information(I) with smart=true (diagonal
    information -> Diagonal).  Missing 2-D vertices are created by chaining odometry (:541-546)."""
    from .graph import noiseModel
    vid, vstate, ei, ej, meas, infos = [], [], [], [], [], []
    for line in open(path):
        t = line.split()
        if not t:
            continue
        tag = t[0]
        if not is3D and tag in ("VERTEX_SE2", "VERTEX2"):
            vid.append(int(t[1]))
            vstate.append([float(t[2]), float(t[3]), float(t[4])])
        elif not is3D and tag in ("EDGE_SE2", "EDGE2", "EDGE", "ODOMETRY"):
            ei.append(int(t[1]))
            ej.append(int(t[2]))
            meas.append([float(t[3]), float(t[4]), float(t[5])])
            v = [float(x) for x in t[6:12]]
            infos.append(np.array([[v[0], v[1], v[2]], [v[1], v[3], v[4]], [v[2], v[4], v[5]]]))
        elif is3D and tag == "VERTEX_SE3:QUAT":
            x, y, z, qx, qy, qz, qw = (float(v) for v in t[2:9])
            vid.append(int(t[1]))
            vstate.append(np.concatenate([Rot3.Quaternion(qw, qx, qy, qz).R.reshape(9), [x, y, z]]))
        elif is3D and tag == "EDGE_SE3:QUAT":
            x, y, z, qx, qy, qz, qw = (float(v) for v in t[3:10])
            ei.append(int(t[1]))
            ej.append(int(t[2]))
            meas.append(np.concatenate([Rot3.Quaternion(qw, qx, qy, qz).R.reshape(9), [x, y, z]]))
            up = [float(v) for v in t[10:31]]
            m = np.zeros((6, 6))
            m[np.triu_indices(6)] = up
            m = m + m.T - np.diag(np.diagonal(m))
            mg = np.zeros((6, 6))
            mg[:3, :3] = m[3:, 3:]
            mg[3:, 3:] = m[:3, :3]
            mg[:3, 3:] = m[3:, :3]
            mg[3:, :3] = m[:3, 3:]
            infos.append(mg)
    ei, ej = np.array(ei, np.int64), np.array(ej, np.int64)
    meas = np.array(meas, dtype=float)
    d = 6 if is3D else 3
    states = {i: np.asarray(s, float) for i, s in zip(vid, vstate)}
    if not is3D:
        for a, b, z in zip(ei, ej, meas):  # chain odometry for missing vertices
            if a not in states:
                states[int(a)] = np.zeros(3)
            if b not in states:
                xa = states[int(a)]
                c, s = math.cos(xa[2]), math.sin(xa[2])
                states[int(b)] = np.array([xa[0] + c * z[0] - s * z[1], xa[1] + s * z[0] + c * z[1], xa[2] + z[2]])
    ids = np.array(sorted(states), np.int64)
    index = {int(k): i for i, k in enumerate(ids)}
    S = np.stack([states[int(k)] for k in ids])
    Rs = []
    for I in infos:
        nm = noiseModel.Gaussian.Information(I)
        if nm.kind == A.NOISE_GAUSSIAN:
            Rs.append(nm.params.reshape(d, d))
        elif nm.kind == A.NOISE_DIAGONAL:
            Rs.append(np.diag(1.0 / nm.params))
        elif nm.kind == A.NOISE_ISOTROPIC:
            Rs.append(np.eye(d) / nm.params[0])
        else:
            Rs.append(np.eye(d))
    prior_sig = np.sqrt([1e-6, 1e-6, 1e-8]) if not is3D else np.sqrt([1e-6] * 3 + [1e-4] * 3)
    arr = _pose_graph_arrays(A.VAR_POSE3 if is3D else A.VAR_POSE2, S,
                             np.array([index[int(a)] for a in ei]), np.array([index[int(b)] for b in ej]),
                             meas, None, prior_sig, keys=ids.astype(np.uint64), noise_R=np.array(Rs))
    if not add_prior:
        raise NotImplementedError("graphs without the anchoring prior are not used on this path")
    return arr


def write_g2o(path: str, arrays: A.ProblemArrays, packed_values: np.ndarray):
    """writeG2o (gtsam/slam/dataset.cpp:636-735) for Pose2 / Pose3 graphs with Diagonal or Gaussian noise."""
    so = arrays.state_offsets()
    _f = lambda x: repr(float(x))
    with open(path, "w") as fh:
        for i, k in enumerate(arrays.var_keys):
            s = packed_values[so[i]:so[i + 1]]
            if arrays.var_types[i] == A.VAR_POSE2:
                fh.write(f"VERTEX_SE2 {int(k)} {_f(s[0])} {_f(s[1])} {_f(s[2])}\n")
            elif arrays.var_types[i] == A.VAR_POSE3:
                qw, qx, qy, qz = _quat_from_R(s[:9].reshape(3, 3))
                fh.write(f"VERTEX_SE3:QUAT {int(k)} {_f(s[9])} {_f(s[10])} {_f(s[11])} {_f(qx)} {_f(qy)} {_f(qz)} {_f(qw)}\n")
        for f in range(arrays.n_factors):
            if arrays.f_type[f] != A.F_BETWEEN:
                continue
            a, b = arrays.f_vars[arrays.f_key_ptr[f]:arrays.f_key_ptr[f + 1]]
            z = arrays.meas[arrays.f_meas_ptr[f]:arrays.f_meas_ptr[f + 1]]
            nz = arrays.noise[arrays.f_noise_ptr[f]:arrays.f_noise_ptr[f + 1]]
            d = int(arrays.f_rows[f])
            if arrays.f_noise_kind[f] == A.NOISE_GAUSSIAN:
                Rm = nz.reshape(d, d)
                info = Rm.T @ Rm
            elif arrays.f_noise_kind[f] == A.NOISE_DIAGONAL:
                info = np.diag(1.0 / nz ** 2)
            elif arrays.f_noise_kind[f] == A.NOISE_ISOTROPIC:
                info = np.eye(d) / nz[0] ** 2
            else:
                info = np.eye(d)
            ka, kb = int(arrays.var_keys[a]), int(arrays.var_keys[b])
            if d == 3:
                up = [info[0, 0], info[0, 1], info[0, 2], info[1, 1], info[1, 2], info[2, 2]]
                fh.write(f"EDGE_SE2 {ka} {kb} {_f(z[0])} {_f(z[1])} {_f(z[2])} " + " ".join(repr(float(v)) for v in up) + "\n")
            else:
                qw, qx, qy, qz = _quat_from_R(z[:9].reshape(3, 3))
                m = np.zeros((6, 6))
                m[:3, :3], m[3:, 3:] = info[3:, 3:], info[:3, :3]
                m[:3, 3:], m[3:, :3] = info[3:, :3], info[:3, 3:]
                up = m[np.triu_indices(6)]
                fh.write(f"EDGE_SE3:QUAT {ka} {kb} {_f(z[9])} {_f(z[10])} {_f(z[11])} {_f(qx)} {_f(qy)} {_f(qz)} {_f(qw)} "
                         + " ".join(repr(float(v)) for v in up) + "\n")


def _quat_from_R(R):
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        return 0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s
    i = int(np.argmax(np.diagonal(R)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = math.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
    q = [0.0] * 4
    q[0] = (R[k, j] - R[j, k]) / s
    q[1 + i] = 0.25 * s
    q[1 + j] = (R[j, i] + R[i, j]) / s
    q[1 + k] = (R[k, i] + R[i, k]) / s
    return tuple(q)


# ------------------------------------------------------------------------------------------
# visual SLAM (GenericProjectionFactor<Pose3,Point3,Cal3_S2>) at scale, and its growth by one keyframe
# ------------------------------------------------------------------------------------------
def synth_visual_slam(n_poses: int, n_points: int, n_obs: int, seed: int = 9, upto: int = None):
    """A vehicle driving along a street (examples/VisualISAM2Example.cpp, SURVEY §8 config 5, scaled): keyframe i at
    (i, 0, 0) looking sideways (+y), landmarks 6-14 m to the side, each seen from a window of k >= 2 consecutive keyframes
    centred on it (sum k = n_obs) — parallax does not shrink as the trajectory grows.  Fixed Cal3_S2 calibration
    (f = 800), pixel sigma 0.5; a prior on the first pose and odometry BetweenFactors along the trajectory.
    `upto` = only the keyframes [0, upto) and the landmarks seen at least twice from them (a landmark enters the graph at
    its second sighting).  Factor order: prior, odometry, then the projections sorted by (keyframe, landmark);
    `factor_ids` (one 64-bit id per factor, stable across `upto`) lets a caller build gsx_update's factor_origin.
    Returns (ProblemArrays, factor_ids)."""
    rng = np.random.default_rng(seed)
    upto = n_poses if upto is None else int(upto)
    f = 800.0
    k = np.full(n_points, 2, np.int64)
    extra = n_obs - 2 * n_points
    if extra < 0:
        raise ValueError("need n_obs >= 2 n_points")
    cap = min(n_poses, 24)
    while extra > 0:
        add = rng.multinomial(extra, np.full(n_points, 1.0 / n_points))
        newk = np.minimum(k + add, cap)
        extra -= int((newk - k).sum())
        k = newk
        if np.all(k == cap):
            raise ValueError("too many observations for this many keyframes")
    start = (rng.random(n_points) * (n_poses - k + 1)).astype(np.int64)
    pt = np.repeat(np.arange(n_points), k)
    cam = np.repeat(start, k) + (np.arange(pt.size) - np.repeat(_csr(k)[:-1], k))
    points = np.stack([start + 0.5 * (k - 1) + rng.uniform(-0.5, 0.5, n_points), rng.uniform(6, 14, n_points),
                       rng.uniform(-2, 2, n_points)], axis=1)
    R1 = np.array([[1.0, 0.0, 0.0], [0.0, 0.0, 1.0], [0.0, -1.0, 0.0]])   # columns: x_c = +x, y_c = -z, z_c = +y (world)
    R = np.broadcast_to(R1, (n_poses, 3, 3)).copy()
    t = np.stack([np.arange(n_poses, dtype=float), np.zeros(n_poses), np.zeros(n_poses)], axis=1)
    q = np.einsum("nji,nj->ni", R[cam], points[pt] - t[cam])
    assert np.all(q[:, 2] > 0)
    uv = f * q[:, :2] / q[:, 2:3] + rng.normal(0, 0.5, (cam.size, 2))
    dR, dt = _pose3_expmap(rng.normal(0, 0.01, (n_poses, 6)))
    init_R = R @ dR
    init_t = t + np.einsum("nij,nj->ni", R, dt)
    init_pts = points + rng.normal(0, 0.05, points.shape)
    keep = cam < upto
    cnt = np.bincount(pt[keep], minlength=n_points)
    keep &= cnt[pt] >= 2
    o = np.lexsort((pt[keep], cam[keep]))
    cam_k, pt_k, uv_k = cam[keep][o], pt[keep][o], uv[keep][o]
    lm = np.flatnonzero(cnt >= 2)
    lm_pos = np.full(n_points, -1, np.int64)
    lm_pos[lm] = np.arange(lm.size)
    n_l, n_x = lm.size, upto
    keys = np.concatenate([(np.uint64(ord("l")) << np.uint64(56)) | lm.astype(np.uint64),
                           (np.uint64(ord("x")) << np.uint64(56)) | np.arange(n_x, dtype=np.uint64)])
    types = np.concatenate([np.full(n_l, A.VAR_VECTOR), np.full(n_x, A.VAR_POSE3)])
    dims = np.concatenate([np.full(n_l, 3), np.full(n_x, 6)])
    # factors: prior on x0, odometry x_{i-1} -> x_i (true relative pose), projections
    Rt = np.concatenate([R.reshape(-1, 9), t], axis=1)[:upto]   # (the prior and the odometry carry the true poses)
    n_odo, n_proj = upto - 1, cam_k.size
    Rrel = np.einsum("nji,njk->nik", R[:upto - 1], R[1:upto])
    trel = np.einsum("nji,nj->ni", R[:upto - 1], t[1:upto] - t[:upto - 1])
    odo_meas = np.concatenate([Rrel.reshape(-1, 9), trel], axis=1)
    nf = 1 + n_odo + n_proj
    f_type = np.concatenate([[A.F_PRIOR], np.full(n_odo, A.F_BETWEEN), np.full(n_proj, A.F_PROJECTION)])
    f_rows = np.concatenate([[6], np.full(n_odo, 6), np.full(n_proj, 2)])
    nk = np.concatenate([[1], np.full(n_odo, 2), np.full(n_proj, 2)])
    xi = n_l + np.arange(upto)
    fv = np.concatenate([[xi[0]], np.stack([xi[:-1], xi[1:]], axis=1).reshape(-1),
                         np.stack([n_l + cam_k, lm_pos[pt_k]], axis=1).reshape(-1)])
    Kv = np.array([f, f, 0.0, 0.0, 0.0])
    meas = np.concatenate([Rt[0], odo_meas.reshape(-1),
                           np.concatenate([uv_k, np.broadcast_to(Kv, (n_proj, 5))], axis=1).reshape(-1)])
    nmeas = np.concatenate([[12], np.full(n_odo, 12), np.full(n_proj, 7)])
    nkind = np.concatenate([[A.NOISE_DIAGONAL], np.full(n_odo, A.NOISE_DIAGONAL), np.full(n_proj, A.NOISE_ISOTROPIC)])
    nnoise = np.concatenate([[6], np.full(n_odo, 6), np.full(n_proj, 1)])
    noise = np.concatenate([[0.01] * 3 + [0.05] * 3, np.tile([0.02] * 3 + [0.1] * 3, n_odo), np.full(n_proj, 0.5)])
    values = np.concatenate([init_pts[lm].reshape(-1),
                             np.concatenate([init_R[:upto].reshape(-1, 9), init_t[:upto]], axis=1).reshape(-1)])
    ids = np.concatenate([[0], 1 + np.arange(n_odo), (1 << 40) + cam_k * np.int64(n_points) + pt_k]).astype(np.int64)
    arr = A.ProblemArrays(
        var_keys=keys, var_types=types, var_dims=dims, f_type=f_type, f_rows=f_rows, f_key_ptr=_csr(nk), f_vars=fv,
        f_meas_ptr=_csr(nmeas), meas=meas, f_noise_kind=nkind, f_noise_ptr=_csr(nnoise), noise=noise, values=values,
        meta=dict(kind="visual_slam", n_poses=upto, n_landmarks=int(n_l), n_obs=int(n_proj), seed=seed))
    return arr, ids
