"""Seeded synthetic maps for the ChainBundle hot path.

The reference ships no datasets or loaders; this generator produces the problems named in
BASELINE.json (multi-camera rig on a loop trajectory, points in a shell around it, noisy
and partly gross-outlier measurements, perturbed initial state) and replays them into any
object that offers the ChainBundle surface (AddPose / AddPoint / AddMeas) in the order
BundleAdjusterMulti::BundleAdjust / BundleAdjusterSingle::BundleAdjust populate it
(/root/reference/src/BundleAdjusterMulti.cc:83-200, BundleAdjusterSingle.cc:76-151).
"""
import math
from dataclasses import dataclass, field

import numpy as np

from .taylor_camera import TaylorCamera

DEFAULT_SEED = 20260927
# (a0, a2, a3, a4, xc, yc, c, d, e): fisheye looking along +z, ~163 deg field of view
DEFAULT_CAM_PARAMS = (250.0, -1.2e-3, 1.0e-7, -1.0e-10, 320.0, 240.0, 1.0, 0.0, 0.0)


def so3_exp(w):
    """TooN SO3::exp (Rodrigues with the TooN small-angle thresholds)."""
    w = np.asarray(w, dtype=np.float64)
    th2 = float(w @ w)
    th = math.sqrt(th2)
    if th2 < 1e-8:
        A, B = 1.0 - th2 / 6.0, 0.5
    elif th2 < 1e-6:
        B = 0.5 - 0.25 * th2 / 6.0
        A = 1.0 - th2 / 6.0 * (1.0 - th2 / 20.0)
    else:
        A, B = math.sin(th) / th, (1 - math.cos(th)) / th2
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    return np.eye(3) + A * K + B * (K @ K)


def se3_exp(mu):
    """TooN SE3::exp, mu = (t, w)."""
    mu = np.asarray(mu, dtype=np.float64)
    t, w = mu[:3], mu[3:]
    th2 = float(w @ w)
    th = math.sqrt(th2)
    cr = np.cross(w, t)
    if th2 < 1e-8:
        tr = t + 0.5 * cr
    else:
        if th2 < 1e-6:
            C = (1.0 - th2 / 20.0) / 6.0
            B = 0.5 - 0.25 * th2 / 6.0
        else:
            A = math.sin(th) / th
            B = (1 - math.cos(th)) / th2
            C = (1 - A) / th2
        tr = t + B * cr + C * np.cross(w, cr)
    return so3_exp(w), tr


def rot_z(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


@dataclass
class Problem:
    """A bundle problem in reference vocabulary, ready to replay into a ChainBundle."""
    cams: list                      # TaylorCamera per camera index
    mode: str                       # "multi" (chains MKF->cam) or "single" (chains KF)
    n_mkf: int
    base_R: np.ndarray              # (P,3,3) BaseFromWorld rotation (initial, perturbed)
    base_t: np.ndarray              # (P,3)
    base_fixed: np.ndarray          # (P,) bool
    cam_R: np.ndarray               # (C,3,3) CamFromBase
    cam_t: np.ndarray               # (C,3)
    pt_x: np.ndarray                # (N,3) point in its source camera frame (initial)
    pt_src: np.ndarray              # (N,2) source (mkf, cam)
    pt_fixed: np.ndarray            # (N,) bool (fixed points are given in world coordinates)
    ms_mkf: np.ndarray              # (M,) observer mkf
    ms_cam: np.ndarray              # (M,)
    ms_pt: np.ndarray               # (M,) point index
    ms_uv: np.ndarray               # (M,2)
    ms_level: np.ndarray            # (M,)
    true_base_R: np.ndarray = None
    true_base_t: np.ndarray = None
    true_world: np.ndarray = None   # (N,3)
    rel_R: np.ndarray = None        # (C,3,3) calib mode: Cam_c-from-Cam_0 (initial, perturbed); entry 0 is the identity
    rel_t: np.ndarray = None        # (C,3)
    ids: dict = field(default_factory=dict)

    @property
    def n_points(self):
        return self.pt_x.shape[0]

    @property
    def n_meas(self):
        return self.ms_uv.shape[0]

    def populate(self, bundle, batch=True):
        """Replay into `bundle` (AddPose/AddPoint/AddMeas or their *_batch forms).

        Returns dict with the bundle ids: 'mkf' (P,), 'cam' (C,), 'point' (N,), 'world'."""
        P, C = self.n_mkf, len(self.cams)
        mkf_id = np.zeros(P, dtype=np.int32)
        cam_id = np.zeros(C, dtype=np.int32)
        world_id = -1
        if self.mode == "multi":
            # BundleAdjusterMulti.cc:83-134: MKF pose, then a fixed pose per camera name when first met
            for k in range(P):
                mkf_id[k] = bundle.AddPose(self.base_R[k], self.base_t[k], bool(self.base_fixed[k]))
                if k == 0:
                    for c in range(C):
                        cam_id[c] = bundle.AddPose(self.cam_R[c], self.cam_t[c], True)
        elif self.mode == "calib":
            # BundleAdjusterCalib.cc:118-146: the free relative camera poses first (the first camera is the identity
            # and is skipped), then one pose per MKF = the first KeyFrame's CamFromWorld
            for c in range(1, C):
                cam_id[c] = bundle.AddPose(self.rel_R[c], self.rel_t[c], False)
            for k in range(P):
                R = self.cam_R[0] @ self.base_R[k]
                t = self.cam_R[0] @ self.base_t[k] + self.cam_t[0]
                mkf_id[k] = bundle.AddPose(R, t, bool(self.base_fixed[k]))
        else:
            # BundleAdjusterSingle.cc:76-101: one pose per KeyFrame = CamFromWorld
            assert C == 1
            for k in range(P):
                R = self.cam_R[0] @ self.base_R[k]
                t = self.cam_R[0] @ self.base_t[k] + self.cam_t[0]
                mkf_id[k] = bundle.AddPose(R, t, bool(self.base_fixed[k]))
        if self.pt_fixed.any():
            world_id = bundle.AddPose(np.eye(3), np.zeros(3), True)      # BundleAdjusterMulti.cc:147-149
        N = self.n_points
        chains = np.zeros((N, 2), dtype=np.int32)
        chain_len = np.zeros(N, dtype=np.int32)
        fx = np.asarray(self.pt_fixed, dtype=bool)
        if N:
            if self.mode == "multi":
                chains[:, 0] = mkf_id[self.pt_src[:, 0]]
                chains[:, 1] = cam_id[self.pt_src[:, 1]]
                chain_len[:] = 2
            elif self.mode == "calib":
                # BundleAdjusterCalib.cc:169-173: chain {MKF} for the first camera, {MKF, relative pose} otherwise
                chains[:, 0] = mkf_id[self.pt_src[:, 0]]
                chain_len[:] = 1
                rel = self.pt_src[:, 1] > 0
                chains[rel, 1] = cam_id[self.pt_src[rel, 1]]
                chain_len[rel] = 2
            else:
                chains[:, 0] = mkf_id[self.pt_src[:, 0]]
                chain_len[:] = 1
            chains[fx, 0] = world_id
            chains[fx, 1] = 0
            chain_len[fx] = 1
        if batch and hasattr(bundle, "AddPointBatch"):
            pt_id = bundle.AddPointBatch(self.pt_x, chains, chain_len, self.pt_fixed)
        else:
            pt_id = np.array([bundle.AddPoint(self.pt_x[i], [int(v) for v in chains[i, :chain_len[i]]], bool(self.pt_fixed[i]))
                              for i in range(N)], dtype=np.int32)
        M = self.n_meas
        mch = np.zeros((M, 2), dtype=np.int32)
        mlen = np.full(M, 2 if self.mode == "multi" else 1, dtype=np.int32)
        mch[:, 0] = mkf_id[self.ms_mkf]
        if self.mode == "multi":
            mch[:, 1] = cam_id[self.ms_cam]
        elif self.mode == "calib":              # BundleAdjusterCalib.cc:190-196
            rel = self.ms_cam > 0
            mch[rel, 1] = cam_id[self.ms_cam[rel]]
            mlen[rel] = 2
        sig = (4.0 ** self.ms_level).astype(np.float64)     # LevelScale^2, BundleAdjusterMulti.cc:196
        if batch and hasattr(bundle, "AddMeasBatch"):
            bundle.AddMeasBatch(mch, mlen, pt_id[self.ms_pt], self.ms_uv, sig, self.ms_cam.astype(np.int32))
        else:
            for j in range(M):
                bundle.AddMeas([int(v) for v in mch[j, :mlen[j]]], int(pt_id[self.ms_pt[j]]), self.ms_uv[j], float(sig[j]), int(self.ms_cam[j]))
        self.ids = {"mkf": mkf_id, "cam": cam_id, "point": np.asarray(pt_id, dtype=np.int32), "world": world_id}
        return self.ids


def make_rig(n_cams, lever=0.1):
    """n_cams cameras at 360/n yaw steps with `lever` m lever arms; camera looks along its +z."""
    R0 = np.array([[0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0]])   # base x-forward/z-up -> cam z-forward/y-down
    Rs, ts = [], []
    for k in range(n_cams):
        yaw = 2 * math.pi * k / max(n_cams, 1) if n_cams > 1 else 0.0
        R = R0 @ rot_z(-yaw)
        p = lever * np.array([math.cos(yaw), math.sin(yaw), 0.0]) if n_cams > 1 else np.zeros(3)
        Rs.append(R)
        ts.append(-R @ p)
    return np.array(Rs), np.array(ts)


def make_problem(n_cams=4, n_mkf=50, n_points=10000, per_point=8, mode="multi", seed=DEFAULT_SEED,
                 radius=5.0, arc_step=None, pixel_sigma=0.5, outlier_frac=0.02, pose_sigma=(0.02, 0.5),
                 depth_sigma=0.05, n_fixed_points=0, cam_params=DEFAULT_CAM_PARAMS, image_size=(640, 480),
                 noise=True, perturb=True, k_near=12, n_fixed_mkf=1, shard=0, newton_camera=False):
    """SURVEY.md 8(d) generator.  Returns a Problem with exactly n_points points and
    per_point*n_points measurements.

    `shard` selects an independent set of points/measurements over the SAME trajectory and the
    same initial pose perturbation (multi-GPU runs: every rank holds all poses and its own shard
    of the map)."""
    rng = np.random.default_rng([seed, 1 + shard])      # points, measurements, noise
    rng_pose = np.random.default_rng([seed, 0])        # pose perturbation: identical on every shard
    # newton_camera: the camera model without a usable inverse polynomial (linear inverse + Newton, TaylorCamera.cc:159-176)
    cam = TaylorCamera(cam_params, image_size, image_size, image_size, force_newton=newton_camera)
    cams = [cam] * n_cams
    cam_R, cam_t = make_rig(n_cams)
    # trajectory: loop (or arc) of radius `radius` with sinusoidal height
    dphi = (2 * math.pi / n_mkf) if arc_step is None else arc_step / radius
    phis = dphi * np.arange(n_mkf)
    centres = np.stack([radius * np.cos(phis), radius * np.sin(phis), 0.3 * np.sin(3 * phis)], axis=1)
    tR = np.array([rot_z(ph + math.pi / 2).T for ph in phis])            # BaseFromWorld rotation
    tt = np.array([-tR[k] @ centres[k] for k in range(n_mkf)])
    # points in a 2..15 m shell around the trajectory; over-generate then keep well-observed ones
    need = n_points
    world = []
    obs_mk, obs_cam, obs_uv = [], [], []                   # per accepted point: per_point observations (MKF, camera, pixel), in candidate order
    kk = min(k_near, n_mkf)
    rounds = 0
    while need > 0:
        rounds += 1
        if rounds > 200:
            raise ValueError("synth.make_problem: cannot find points with %d valid observations "
                             "(n_mkf=%d, n_cams=%d): lower per_point" % (per_point, n_mkf, n_cams))
        n_try = int(need * 1.5) + 64
        anchor = rng.integers(0, n_mkf, n_try)
        u = rng.normal(size=(n_try, 3))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        d = rng.uniform(2.0, 15.0, n_try)
        X = centres[anchor] + u * d[:, None]
        # the kk nearest MKF centres of every candidate, nearest first (ties by index, as a stable argsort of the whole row gives them);
        # in row chunks: the (n_try, P, 3) difference array of a 100k-point round would be 1.8 GB
        near = np.empty((n_try, kk), dtype=np.int64)
        for r0 in range(0, n_try, 8192):
            d2 = ((X[r0:r0 + 8192, None, :] - centres[None, :, :]) ** 2).sum(axis=2)   # (chunk, P)
            if kk < n_mkf:
                part = np.argpartition(d2, kk - 1, axis=1)[:, :kk]
                pd = np.take_along_axis(d2, part, axis=1)
                o2 = np.lexsort((part, pd), axis=1)                  # by distance, then by index
                near[r0:r0 + 8192] = np.take_along_axis(part, o2, axis=1)
            else:
                near[r0:r0 + 8192] = np.argsort(d2, axis=1, kind="stable")[:, :kk]
        # candidate observations ordered by (distance rank, cam)
        cand_valid = np.zeros((n_try, kk, n_cams), dtype=bool)
        cand_uv = np.zeros((n_try, kk, n_cams, 2))
        for r in range(kk):
            mk = near[:, r]
            xb = np.einsum("nij,nj->ni", tR[mk], X) + tt[mk]
            for c in range(n_cams):
                xc = xb @ cam_R[c].T + cam_t[c]
                uv, inv = cam.project(xc)
                cand_valid[:, r, c] = ~inv
                cand_uv[:, r, c] = uv
        flat_valid = cand_valid.reshape(n_try, -1)
        count = flat_valid.sum(axis=1)
        good = np.nonzero(count >= per_point)[0][:need]
        if len(good):
            # the first per_point valid candidates of every accepted point, in candidate order (distance rank, then camera)
            fv = flat_valid[good]
            pick = fv & (np.cumsum(fv, axis=1) <= per_point)
            gi, si = np.nonzero(pick)                          # row-major: per point, ascending candidate index
            rank, camc = si // n_cams, si % n_cams
            obs_mk.append(near[good[gi], rank].reshape(len(good), per_point))
            obs_cam.append(camc.reshape(len(good), per_point))
            obs_uv.append(cand_uv[good[gi], rank, camc].reshape(len(good), per_point, 2))
            world.append(X[good])
        need -= len(good)
    world = np.concatenate(world) if world else np.zeros((0, 3))
    obs_mk = np.concatenate(obs_mk).astype(np.int64); obs_cam = np.concatenate(obs_cam).astype(np.int64); obs_uv = np.concatenate(obs_uv)
    N = n_points
    # fixed (calibration) points: stored in world coordinates with chain {world}
    pt_fixed = np.zeros(N, dtype=bool)
    if n_fixed_points:
        pt_fixed[rng.choice(N, n_fixed_points, replace=False)] = True
    # measurements, populated MKF-major then camera then point (BundleAdjusterMulti.cc:168-200)
    pt_src = np.zeros((N, 2), dtype=np.int32)
    first = np.argmin(obs_mk*n_cams + obs_cam, axis=1)                    # "first KF that sees it": smallest (MKF, camera)
    pt_src[:, 0] = obs_mk[np.arange(N), first]; pt_src[:, 1] = obs_cam[np.arange(N), first]
    pt_of = np.repeat(np.arange(N), per_point)
    order = np.lexsort((pt_of, obs_cam.reshape(-1), obs_mk.reshape(-1)))   # by (MKF, camera, point)
    ms_mkf = obs_mk.reshape(-1)[order].astype(np.int32)
    ms_cam = obs_cam.reshape(-1)[order].astype(np.int32)
    ms_pt = pt_of[order].astype(np.int32)
    ms_uv = np.ascontiguousarray(obs_uv.reshape(-1, 2)[order], dtype=np.float64)
    M = ms_uv.shape[0]
    ms_level = rng.choice(4, size=M, p=[0.55, 0.25, 0.15, 0.05]).astype(np.int32)
    if noise:
        ms_uv = ms_uv + rng.normal(size=(M, 2)) * (pixel_sigma * (2.0 ** ms_level))[:, None]
        n_out = int(round(outlier_frac * M))
        if n_out:
            oi = rng.choice(M, n_out, replace=False)
            ms_uv[oi] = rng.uniform([0, 0], image_size, size=(n_out, 2))
    # true relative coordinates in the (true) source camera frame
    srcR = np.einsum("nij,njk->nik", cam_R[pt_src[:, 1]], tR[pt_src[:, 0]])
    srct = np.einsum("nij,nj->ni", cam_R[pt_src[:, 1]], tt[pt_src[:, 0]]) + cam_t[pt_src[:, 1]]
    x_rel = np.einsum("nij,nj->ni", srcR, world) + srct
    base_R, base_t = tR.copy(), tt.copy()
    base_fixed = np.zeros(n_mkf, dtype=bool)
    base_fixed[:n_fixed_mkf] = True
    pt_x = x_rel.copy()
    if perturb:
        for k in range(n_mkf):
            if base_fixed[k]:
                continue
            xi = np.concatenate([rng_pose.normal(size=3) * pose_sigma[0], rng_pose.normal(size=3) * math.radians(pose_sigma[1])])
            R, t = se3_exp(xi)
            base_R[k] = R @ tR[k]
            base_t[k] = R @ tt[k] + t
        pt_x = x_rel * (1.0 + rng.normal(size=(N, 1)) * depth_sigma)
    pt_x[pt_fixed] = world[pt_fixed]
    rel_R = rel_t = None
    if mode == "calib":
        # relative camera poses Cam_c-from-Cam_0 = CamFromBase_c * CamFromBase_0^-1, the unknowns of the calibration
        rel_R = np.array([cam_R[c] @ cam_R[0].T for c in range(n_cams)])
        rel_t = np.array([cam_t[c] - rel_R[c] @ cam_t[0] for c in range(n_cams)])
        if perturb:
            for c in range(1, n_cams):
                xi = np.concatenate([rng_pose.normal(size=3) * pose_sigma[0], rng_pose.normal(size=3) * math.radians(pose_sigma[1])])
                R, t = se3_exp(xi)
                rel_R[c], rel_t[c] = R @ rel_R[c], R @ rel_t[c] + t
    return Problem(rel_R=rel_R, rel_t=rel_t, cams=cams, mode=mode, n_mkf=n_mkf, base_R=base_R, base_t=base_t, base_fixed=base_fixed,
                   cam_R=cam_R, cam_t=cam_t, pt_x=pt_x, pt_src=pt_src, pt_fixed=pt_fixed,
                   ms_mkf=ms_mkf, ms_cam=ms_cam, ms_pt=ms_pt, ms_uv=ms_uv, ms_level=ms_level,
                   true_base_R=tR, true_base_t=tt, true_world=world)


def merge_shards(shards):
    """One Problem holding the points and measurements of all shards (same trajectory): the single-rank
    equivalent of a sharded multi-GPU run."""
    a = shards[0]
    off, pt_x, pt_src, pt_fixed, ms_mkf, ms_cam, ms_pt, ms_uv, ms_level, tw = 0, [], [], [], [], [], [], [], [], []
    for s in shards:
        assert np.array_equal(s.base_R, a.base_R) and np.array_equal(s.base_t, a.base_t)
        pt_x.append(s.pt_x); pt_src.append(s.pt_src); pt_fixed.append(s.pt_fixed)
        ms_mkf.append(s.ms_mkf); ms_cam.append(s.ms_cam); ms_pt.append(s.ms_pt + off)
        ms_uv.append(s.ms_uv); ms_level.append(s.ms_level); tw.append(s.true_world)
        off += s.n_points
    return Problem(cams=a.cams, mode=a.mode, n_mkf=a.n_mkf, base_R=a.base_R.copy(), base_t=a.base_t.copy(), base_fixed=a.base_fixed,
                   cam_R=a.cam_R, cam_t=a.cam_t, pt_x=np.concatenate(pt_x), pt_src=np.concatenate(pt_src),
                   pt_fixed=np.concatenate(pt_fixed), ms_mkf=np.concatenate(ms_mkf), ms_cam=np.concatenate(ms_cam),
                   ms_pt=np.concatenate(ms_pt), ms_uv=np.concatenate(ms_uv), ms_level=np.concatenate(ms_level),
                   true_base_R=a.true_base_R, true_base_t=a.true_base_t, true_world=np.concatenate(tw))


def partition(p, world, rank=None):
    """Split ONE map over `world` ranks the way SURVEY.md 8(e) prescribes: points sorted by the MKF they are expressed in, cut into
    contiguous blocks that balance the MEASUREMENT counts; every measurement lives with its point's owner; poses, cameras and the
    fixed flags are replicated.  What BundleAdjusterMulti hands over is one population (src/BundleAdjusterMulti.cc:90-200) -- this is
    the step that turns it into the per-rank shards the sharded solve takes (BASELINE c4: one 800k-measurement map over 8 GPUs).
    Returns the list of shard Problems (or shard `rank` alone); shard.part = dict(points=map point indices in shard order,
    meas=map measurement indices in shard order).  merge_shards() of the result is the map again up to the point order."""
    assert world >= 1
    N, M = p.n_points, p.n_meas
    per_pt = np.bincount(p.ms_pt, minlength=N).astype(np.int64)
    order = np.argsort(p.pt_src[:, 0], kind="stable")              # by source MKF, ties in map order
    cum = np.cumsum(per_pt[order])
    # block b ends where the running measurement count first reaches (b + 1) M / world
    cuts = [0] + [int(np.searchsorted(cum, (b + 1) * M / world, side="left")) + 1 for b in range(world - 1)] + [N]
    cuts = [min(max(c, 0), N) for c in cuts]
    for b in range(1, len(cuts)):
        cuts[b] = max(cuts[b], cuts[b - 1])
    owner = np.empty(N, dtype=np.int64)
    for b in range(world):
        owner[order[cuts[b]:cuts[b + 1]]] = b
    shards = []
    for b in (range(world) if rank is None else [rank]):
        pts = order[cuts[b]:cuts[b + 1]]
        pmap = -np.ones(N, dtype=np.int64)
        pmap[pts] = np.arange(len(pts))
        ms = np.flatnonzero(owner[p.ms_pt] == b)                      # map order is kept: MKF, camera, point
        q = Problem(cams=p.cams, mode=p.mode, n_mkf=p.n_mkf, base_R=p.base_R.copy(), base_t=p.base_t.copy(), base_fixed=p.base_fixed.copy(),
                    cam_R=p.cam_R, cam_t=p.cam_t, pt_x=p.pt_x[pts].copy(), pt_src=p.pt_src[pts].copy(), pt_fixed=p.pt_fixed[pts].copy(),
                    ms_mkf=p.ms_mkf[ms].copy(), ms_cam=p.ms_cam[ms].copy(), ms_pt=pmap[p.ms_pt[ms]].astype(p.ms_pt.dtype),
                    ms_uv=p.ms_uv[ms].copy(), ms_level=p.ms_level[ms].copy(), true_base_R=p.true_base_R, true_base_t=p.true_base_t,
                    true_world=None if p.true_world is None else p.true_world[pts], rel_R=p.rel_R, rel_t=p.rel_t)
        q.part = dict(points=pts, meas=ms, rank=b, world=world)
        shards.append(q)
    return shards if rank is None else shards[0]


# the BASELINE.json configurations (SURVEY.md 8 notation)
CONFIGS = {
    "c1": dict(n_cams=1, n_mkf=10, n_points=500, per_point=6, mode="single", arc_step=0.3, n_fixed_mkf=2),  # 2 fixed KFs pin the monocular scale gauge
    # calibration-phase shapes (BundleAdjusterCalib): free relative camera poses shared by every MKF, fixed board points
    "calib": dict(n_cams=3, n_mkf=12, n_points=600, per_point=6, mode="calib", arc_step=0.3, n_fixed_points=120,
                  outlier_frac=0.0),      # board corners: no gross outliers (fixed-point measurements are never down-weighted)
    "c2": dict(n_cams=4, n_mkf=50, n_points=10000, per_point=8, mode="multi"),
    "metric": dict(n_cams=4, n_mkf=200, n_points=50000, per_point=8, mode="multi"),
    "c4": dict(n_cams=4, n_mkf=500, n_points=100000, per_point=8, mode="multi"),
    "tiny": dict(n_cams=2, n_mkf=6, n_points=60, per_point=4, mode="multi", arc_step=0.4),
    # not BASELINE configurations: trajectories whose poses see the points of their neighbours only -- a banded reduced system (an open arc)
    # and a cyclic band (a loop walked once) -- the maps the factorisation takes as two chains; `band_metric`: at the headline's sizes
    "band": dict(n_cams=2, n_mkf=130, n_points=6000, per_point=4, mode="multi", radius=120.0, arc_step=2.0, k_near=5),
    "ring": dict(n_cams=2, n_mkf=130, n_points=6000, per_point=4, mode="multi", radius=41.0, k_near=5),
    "band_metric": dict(n_cams=4, n_mkf=200, n_points=50000, per_point=8, mode="multi", radius=400.0, arc_step=4.0, k_near=6),
    "ring_metric": dict(n_cams=4, n_mkf=200, n_points=50000, per_point=8, mode="multi", radius=130.0, k_near=6),
}


def shuffle_mkfs(p, seed=1):
    """The same map with its MKFs handed over in another order (MCPTAM's adapters walk a std::set of MultiKeyFrame pointers,
    /root/reference/src/BundleAdjusterMulti.cc:83-134: address order, not trajectory order): MKF k of the result is MKF perm[k] of `p`,
    measurements again MKF-major then camera then point.  Poses, points and measurements are the same physical quantities."""
    import copy
    rng = np.random.default_rng([DEFAULT_SEED, 77, seed])
    P = p.n_mkf
    perm = rng.permutation(P)
    inv = np.empty(P, dtype=np.int64); inv[perm] = np.arange(P)
    q = copy.copy(p)
    q.base_R = p.base_R[perm]; q.base_t = p.base_t[perm]; q.base_fixed = p.base_fixed[perm]
    if p.true_base_R is not None:
        q.true_base_R = p.true_base_R[perm]; q.true_base_t = p.true_base_t[perm]
    q.pt_src = p.pt_src.copy(); q.pt_src[:, 0] = inv[p.pt_src[:, 0]]
    mk = inv[p.ms_mkf]
    order = np.lexsort((p.ms_pt, p.ms_cam, mk))
    q.ms_mkf = mk[order].astype(p.ms_mkf.dtype); q.ms_cam = p.ms_cam[order]; q.ms_pt = p.ms_pt[order]
    q.ms_uv = np.ascontiguousarray(p.ms_uv[order]); q.ms_level = p.ms_level[order]
    q.ids = {}
    return q


def erase_measurements(p, outliers, ids=None):
    """The map after MapMakerServerBase::HandleOutliers (/root/reference/src/MapMakerServerBase.cc:1198-1238: kf.EraseMeasurementOfPoint for
    every measurement the adjustment flagged): the listed (point id, MKF id, camera index) measurements -- what GetOutlierMeasurements()
    returns -- erased, everything else, the ORDER included, as it was.  `ids`: the id dict populate() returned (default p.ids)."""
    import copy
    ids = p.ids if ids is None else ids
    if not outliers:
        return copy.copy(p)
    P, C = p.n_mkf, len(p.cams)
    pt_of = np.full(int(np.max(ids["point"])) + 1, -1, dtype=np.int64); pt_of[ids["point"]] = np.arange(p.n_points)
    kf_of = np.full(int(np.max(ids["mkf"])) + 1, -1, dtype=np.int64); kf_of[ids["mkf"]] = np.arange(P)
    o = np.asarray(outliers, dtype=np.int64).reshape(-1, 3)
    gone = (pt_of[o[:, 0]] * P + kf_of[o[:, 1]]) * C + o[:, 2]
    key = (p.ms_pt.astype(np.int64) * P + p.ms_mkf.astype(np.int64)) * C + p.ms_cam.astype(np.int64)
    keep = ~np.isin(key, gone)
    q = copy.copy(p)
    q.ms_mkf, q.ms_cam, q.ms_pt, q.ms_uv, q.ms_level = p.ms_mkf[keep], p.ms_cam[keep], p.ms_pt[keep], p.ms_uv[keep], p.ms_level[keep]
    q.ids = {}
    return q


def recent_window(p, n_recent=3, newest=None):
    """The local bundle BundleAdjusterBase::BundleAdjustRecent builds from a map (src/BundleAdjusterBase.cc:188-265): the newest
    MKF and its `n_recent` closest MKFs, of which only the movable ones are kept (snRecentNum, :47), are adjusted; the points are
    those measured from them by at least two KeyFrames; every other MKF that measures one of those points enters FIXED; the
    measurements are all measurements of those points (BundleAdjusterMulti.cc:168-200).  Returns a new multi-mode Problem over
    the MKFs involved (in map order) plus `window` = dict(mkf=map indices kept, adjust=map indices adjusted, points=map point
    indices).  MKF distance is the distance between base origins (the reference takes the closest pair of KeyFrame centres)."""
    assert p.mode == "multi"
    newest = p.n_mkf - 1 if newest is None else int(newest)
    centres = -np.einsum("kji,kj->ki", p.base_R, p.base_t)                     # -R^T t
    d = np.linalg.norm(centres - centres[newest], axis=1)
    d[newest] = np.inf
    closest = np.argsort(d, kind="stable")[:n_recent]
    adjust = {newest} | {int(k) for k in closest if not p.base_fixed[k]}
    adj_mask = np.zeros(p.n_mkf, dtype=bool)
    adj_mask[list(adjust)] = True
    # points seen from the adjusted MKFs, measured by >= 2 KeyFrames overall
    kf_key = p.ms_mkf.astype(np.int64) * len(p.cams) + p.ms_cam
    per_point = np.zeros(p.n_points, dtype=np.int64)
    uniq = np.unique(np.stack([p.ms_pt.astype(np.int64), kf_key], axis=1), axis=0)
    np.add.at(per_point, uniq[:, 0], 1)
    pts_mask = np.zeros(p.n_points, dtype=bool)
    pts_mask[p.ms_pt[adj_mask[p.ms_mkf]]] = True
    pts_mask &= per_point >= 2
    ms_keep = pts_mask[p.ms_pt]
    mkf_keep = np.zeros(p.n_mkf, dtype=bool)
    mkf_keep[p.ms_mkf[ms_keep]] = True
    mkf_keep[p.pt_src[pts_mask, 0]] = True
    mkf_keep |= adj_mask
    kmap = -np.ones(p.n_mkf, dtype=np.int64)
    kmap[mkf_keep] = np.arange(mkf_keep.sum())
    pmap = -np.ones(p.n_points, dtype=np.int64)
    pmap[pts_mask] = np.arange(pts_mask.sum())
    src = p.pt_src[pts_mask].copy()
    src[:, 0] = kmap[src[:, 0]]
    q = Problem(cams=p.cams, mode="multi", n_mkf=int(mkf_keep.sum()), base_R=p.base_R[mkf_keep].copy(), base_t=p.base_t[mkf_keep].copy(),
                base_fixed=~adj_mask[mkf_keep], cam_R=p.cam_R, cam_t=p.cam_t, pt_x=p.pt_x[pts_mask].copy(), pt_src=src.astype(p.pt_src.dtype),
                pt_fixed=p.pt_fixed[pts_mask].copy(), ms_mkf=kmap[p.ms_mkf[ms_keep]].astype(p.ms_mkf.dtype), ms_cam=p.ms_cam[ms_keep].copy(),
                ms_pt=pmap[p.ms_pt[ms_keep]].astype(p.ms_pt.dtype), ms_uv=p.ms_uv[ms_keep].copy(), ms_level=p.ms_level[ms_keep].copy(),
                true_base_R=None if p.true_base_R is None else p.true_base_R[mkf_keep], true_base_t=None if p.true_base_t is None else p.true_base_t[mkf_keep],
                true_world=None if p.true_world is None else p.true_world[pts_mask])
    q.window = dict(mkf=np.flatnonzero(mkf_keep), adjust=np.array(sorted(adjust)), points=np.flatnonzero(pts_mask))
    return q


def make_config(name, **over):
    """BASELINE configuration by name; keyword overrides go to make_problem."""
    kw = dict(CONFIGS[name])
    kw.update(over)
    return make_problem(**kw)
