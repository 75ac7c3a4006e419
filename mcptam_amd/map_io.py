"""Map interchange with MCPTAM's own dump files, so that real maps can be replayed through the HIP back end.

Two text formats, both written by the reference with plain ``ofstream <<`` (default 6 significant digits):

* the map dump, ``MapMakerBase::DumpToFile`` (src/MapMakerBase.cc:475-577): camera poses in the MKF frame, MKF poses in
  the world frame (both stored as the INVERSE of the PTAM-style transform, position + quaternion x,y,z,w), points in
  world coordinates with their parent MKF number / camera name, and the measurements
  ``MKF, camera, point, u, v (level 0), LevelScale(level)^2``;
* the camera dump, ``SystemBase::DumpCamerasToFile`` (src/SystemBase.cc:166-215): name, image size, projection centre,
  polynomial ``a0, 0, a2, a3, a4``, affine ``c, d, e`` and the inverse polynomial (which is re-fitted on load, exactly
  as ``TaylorCamera::RefreshParams`` does from the forward model, src/TaylorCamera.cc:84-198).

``problem_from_map`` turns a loaded map into the population order ``BundleAdjusterMulti`` uses
(src/BundleAdjusterMulti.cc:83-200), ready for ``Problem.populate(ChainBundle)``.
"""
from dataclasses import dataclass, field

import numpy as np

from .synth import Problem
from .taylor_camera import TaylorCamera


# ------------------------------------------------------------------------------------------------ rotations
def quat_from_matrix(R):
    """(x, y, z, w) of a rotation matrix -- tf::Matrix3x3::getRotation's branches (largest of trace / diagonal)."""
    R = np.asarray(R, dtype=np.float64)
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    q = np.zeros(4)
    if tr > 0.0:
        s = np.sqrt(tr + 1.0)
        q[3] = 0.5 * s
        s = 0.5 / s
        q[0] = (R[2, 1] - R[1, 2]) * s
        q[1] = (R[0, 2] - R[2, 0]) * s
        q[2] = (R[1, 0] - R[0, 1]) * s
    else:
        i = 0 if R[0, 0] >= max(R[1, 1], R[2, 2]) else (1 if R[1, 1] >= R[2, 2] else 2)
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[i] = 0.5 * s
        s = 0.5 / s
        q[3] = (R[k, j] - R[j, k]) * s
        q[j] = (R[j, i] + R[i, j]) * s
        q[k] = (R[k, i] + R[i, k]) * s
    return q


def matrix_from_quat(q):
    x, y, z, w = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _inverse(R, t):
    return R.T, -R.T @ t


# ------------------------------------------------------------------------------------------------ containers
@dataclass
class MapFile:
    """The content of one DumpToFile map, in file vocabulary."""
    cam_names: list                  # C camera names, file order (= std::map order of the first MKF's KeyFrames)
    cam_pos: np.ndarray              # (C,3)  position of the camera in the MKF (base) frame  = (CamFromBase)^-1
    cam_quat: np.ndarray             # (C,4)  x,y,z,w
    mkf_pos: np.ndarray              # (P,3)  position of the MKF in the world frame           = (BaseFromWorld)^-1
    mkf_quat: np.ndarray             # (P,4)
    pt_world: np.ndarray             # (N,3)
    pt_parent_mkf: np.ndarray        # (N,)   MKF number of the patch source KeyFrame
    pt_parent_cam: list              # N camera names
    ms_mkf: np.ndarray               # (M,)
    ms_cam: list                     # M camera names
    ms_pt: np.ndarray                # (M,)
    ms_uv: np.ndarray                # (M,2)  v2RootPos, level-0 pixel coordinates
    ms_noise: np.ndarray             # (M,)   LevelScale(level)^2 = 1, 4, 16, 64
    comments: list = field(default_factory=list)


def _fmt(v, precision):
    """``ofstream << double`` with the stream's precision (default 6): %g."""
    return ("%%.%dg" % precision) % v


# ------------------------------------------------------------------------------------------------ map dump
def dump_map(path, m, precision=6):
    """Write `m` in the reference's DumpToFile layout (precision=6 reproduces its default stream formatting; use 17
    for a lossless round trip)."""
    f = lambda v: _fmt(float(v), precision)            # noqa: E731
    with open(path, "w") as o:
        o.write("% Camera poses in MKF frame, format:\n% Total number of cameras\n"
                "% Camera Name, Position (3 vector), Orientation (quaternion, 4 vector)\n")
        o.write("%d\n" % len(m.cam_names))
        for c, name in enumerate(m.cam_names):
            o.write(name + "".join(", " + f(v) for v in list(m.cam_pos[c]) + list(m.cam_quat[c])) + "\n")
        o.write("% MKFs in world frame, format:\n% Total number of MKFs\n"
                "% MKF number, Position (3 vector), Orientation (quaternion, 4 vector)\n")
        o.write("%d\n" % len(m.mkf_pos))
        for k in range(len(m.mkf_pos)):
            o.write("%d" % k + "".join(", " + f(v) for v in list(m.mkf_pos[k]) + list(m.mkf_quat[k])) + "\n")
        o.write("% Points in world frame, format:\n% Total number of points\n"
                "% Point number, Position (3 vector), Parent MKF number, Parent camera name\n")
        o.write("%d\n" % len(m.pt_world))
        for i in range(len(m.pt_world)):
            o.write("%d, %s, %s, %s, %d, %s\n" % (i, f(m.pt_world[i, 0]), f(m.pt_world[i, 1]), f(m.pt_world[i, 2]),
                                                  int(m.pt_parent_mkf[i]), m.pt_parent_cam[i]))
        o.write("% Measurements of points from KeyFrames, format: \n% Total number of measurements\n"
                "% MKF number, camera name, point number, image position (2 vector) at level 0, measurement noise\n")
        o.write("%d\n" % len(m.ms_pt))
        for j in range(len(m.ms_pt)):
            o.write("%d, %s, %d, %s, %s, %s\n" % (int(m.ms_mkf[j]), m.ms_cam[j], int(m.ms_pt[j]), f(m.ms_uv[j, 0]), f(m.ms_uv[j, 1]),
                                                  f(m.ms_noise[j])))
        o.write("% The end")


def _records(path):
    """Non-comment lines split at ', ' (the files have no quoting; names contain no commas)."""
    comments, rows = [], []
    with open(path) as fh:
        for line in fh:
            line = line.rstrip("\n")
            if not line.strip():
                continue
            if line.lstrip().startswith("%"):
                comments.append(line)
                continue
            rows.append([t.strip() for t in line.split(",")])
    return comments, rows


def load_map(path):
    comments, rows = _records(path)
    it = iter(rows)

    def count():
        r = next(it)
        if len(r) != 1:
            raise ValueError("map file %s: expected a count line, got %r" % (path, r))
        return int(r[0])

    nc = count()
    cams = [next(it) for _ in range(nc)]
    nk = count()
    mkfs = [next(it) for _ in range(nk)]
    npt = count()
    pts = [next(it) for _ in range(npt)]
    nm = count()
    ms = [next(it) for _ in range(nm)]
    for name, rec, width in (("camera", cams, 8), ("MKF", mkfs, 8), ("point", pts, 6), ("measurement", ms, 6)):
        for r in rec:
            if len(r) != width:
                raise ValueError("map file %s: malformed %s record %r" % (path, name, r))
    if [int(r[0]) for r in mkfs] != list(range(nk)) or [int(r[0]) for r in pts] != list(range(npt)):
        raise ValueError("map file %s: MKF / point numbers are not consecutive" % path)
    A = lambda rec, a, b: np.array([[float(v) for v in r[a:b]] for r in rec], dtype=np.float64).reshape(len(rec), b - a)  # noqa: E731
    return MapFile(cam_names=[r[0] for r in cams], cam_pos=A(cams, 1, 4), cam_quat=A(cams, 4, 8),
                   mkf_pos=A(mkfs, 1, 4), mkf_quat=A(mkfs, 4, 8),
                   pt_world=A(pts, 1, 4), pt_parent_mkf=np.array([int(r[4]) for r in pts], dtype=np.int32), pt_parent_cam=[r[5] for r in pts],
                   ms_mkf=np.array([int(r[0]) for r in ms], dtype=np.int32), ms_cam=[r[1] for r in ms],
                   ms_pt=np.array([int(r[2]) for r in ms], dtype=np.int32), ms_uv=A(ms, 3, 5), ms_noise=A(ms, 5, 6)[:, 0],
                   comments=comments)


# ------------------------------------------------------------------------------------------------ camera dump
def dump_cameras(path, cameras, precision=6):
    """`cameras`: dict name -> TaylorCamera (written in sorted-name order like the std::map)."""
    f = lambda v: _fmt(float(v), precision)            # noqa: E731
    with open(path, "w") as o:
        o.write("% Camera calibration parameters, format:\n% Total number of cameras\n"
                "% Camera Name, image size (2 vector), projection center (2 vector), polynomial coefficients (5 vector), "
                "affine matrix components (3 vector), inverse polynomial coefficents (variable size)\n")
        o.write("%d\n" % len(cameras))
        for name in sorted(cameras):
            c = cameras[name]
            p = c.params
            vals = [p[4], p[5], p[0], 0, p[1], p[2], p[3], p[6], p[7], p[8]] + list(c.inv_coeffs)
            o.write("%s, %d, %d" % (name, c.image_size[0], c.image_size[1]) + "".join(", " + (f(v) if not isinstance(v, int) else "%d" % v) for v in vals) + "\n")
        o.write("% The end")


def load_cameras(path):
    """dict name -> TaylorCamera.  The stored inverse polynomial lacks its theta normalisation, so it is re-fitted from
    the forward model; the stored coefficients are returned alongside for inspection (`camera.file_inv_poly`)."""
    _, rows = _records(path)
    n = int(rows[0][0])
    out = {}
    for r in rows[1:1 + n]:
        name, w, h = r[0], int(r[1]), int(r[2])
        xc, yc, a0, _a1, a2, a3, a4, c, d, e = [float(v) for v in r[3:13]]
        cam = TaylorCamera([a0, a2, a3, a4, xc, yc, c, d, e], (w, h), (w, h), (w, h))
        cam.file_inv_poly = np.array([float(v) for v in r[13:]])
        out[name] = cam
    return out


# ------------------------------------------------------------------------------------------------ Problem <-> MapFile
def map_from_problem(p, cam_names=None, state=None):
    """MapFile of a synthetic (multi-mode) Problem.  `state` = (base_R, base_t, world) overrides the initial state,
    e.g. with an adjusted one read back from a ChainBundle."""
    assert p.mode == "multi"
    C = len(p.cams)
    names = list(cam_names) if cam_names else ["camera%d" % (c + 1) for c in range(C)]
    bR, bt = (p.base_R, p.base_t) if state is None else (state[0], state[1])
    if state is not None:
        world = state[2]
    else:
        # x_world = (CamFromBase_src * BaseFromWorld_src)^-1 * x_rel
        world = np.zeros((p.n_points, 3))
        for i in range(p.n_points):
            if p.pt_fixed[i]:
                world[i] = p.pt_x[i]
                continue
            k, c = p.pt_src[i]
            R = p.cam_R[c] @ bR[k]
            t = p.cam_R[c] @ bt[k] + p.cam_t[c]
            world[i] = R.T @ (p.pt_x[i] - t)
    cp, cq = np.zeros((C, 3)), np.zeros((C, 4))
    for c in range(C):
        Ri, ti = _inverse(p.cam_R[c], p.cam_t[c])
        cp[c], cq[c] = ti, quat_from_matrix(Ri)
    kp, kq = np.zeros((p.n_mkf, 3)), np.zeros((p.n_mkf, 4))
    for k in range(p.n_mkf):
        Ri, ti = _inverse(bR[k], bt[k])
        kp[k], kq[k] = ti, quat_from_matrix(Ri)
    return MapFile(cam_names=names, cam_pos=cp, cam_quat=cq, mkf_pos=kp, mkf_quat=kq, pt_world=world,
                   pt_parent_mkf=p.pt_src[:, 0].astype(np.int32), pt_parent_cam=[names[c] for c in p.pt_src[:, 1]],
                   ms_mkf=p.ms_mkf.astype(np.int32), ms_cam=[names[c] for c in p.ms_cam], ms_pt=p.ms_pt.astype(np.int32),
                   ms_uv=p.ms_uv.copy(), ms_noise=4.0 ** p.ms_level)


def problem_from_map(m, cameras, n_fixed_mkf=1):
    """A multi-mode Problem from a loaded map.  `cameras`: dict name -> TaylorCamera (load_cameras).  The first
    `n_fixed_mkf` MKFs are fixed (the map's first MKF is the fixed one, include/mcptam/KeyFrame.h:336)."""
    names = list(m.cam_names)
    idx = {n: i for i, n in enumerate(names)}
    missing = [n for n in names if n not in cameras]
    if missing:
        raise ValueError("no calibration for cameras %r" % missing)
    C, P, N = len(names), len(m.mkf_pos), len(m.pt_world)
    cam_R, cam_t = np.zeros((C, 3, 3)), np.zeros((C, 3))
    for c in range(C):
        cam_R[c], cam_t[c] = _inverse(matrix_from_quat(m.cam_quat[c]), m.cam_pos[c])           # CamFromBase
    base_R, base_t = np.zeros((P, 3, 3)), np.zeros((P, 3))
    for k in range(P):
        base_R[k], base_t[k] = _inverse(matrix_from_quat(m.mkf_quat[k]), m.mkf_pos[k])         # BaseFromWorld
    pt_src = np.stack([m.pt_parent_mkf, np.array([idx[n] for n in m.pt_parent_cam], dtype=np.int32)], axis=1).astype(np.int32)
    srcR = np.einsum("nij,njk->nik", cam_R[pt_src[:, 1]], base_R[pt_src[:, 0]])
    srct = np.einsum("nij,nj->ni", cam_R[pt_src[:, 1]], base_t[pt_src[:, 0]]) + cam_t[pt_src[:, 1]]
    pt_x = np.einsum("nij,nj->ni", srcR, m.pt_world) + srct          # BundleAdjusterMulti.cc:153
    level = np.rint(np.log(np.maximum(m.ms_noise, 1.0)) / np.log(4.0)).astype(np.int32)
    if not np.allclose(4.0 ** level, m.ms_noise, rtol=1e-4):
        raise ValueError("measurement noise is not LevelScale(level)^2")
    # population order of BundleAdjusterMulti.cc:168-200: MKF-major, then camera name, then point
    ms_cam = np.array([idx[n] for n in m.ms_cam], dtype=np.int32)
    order = np.lexsort((m.ms_pt, ms_cam, m.ms_mkf))
    fixed = np.zeros(P, dtype=bool)
    fixed[:n_fixed_mkf] = True
    return Problem(cams=[cameras[n] for n in names], mode="multi", n_mkf=P, base_R=base_R, base_t=base_t, base_fixed=fixed,
                   cam_R=cam_R, cam_t=cam_t, pt_x=pt_x, pt_src=pt_src, pt_fixed=np.zeros(N, dtype=bool),
                   ms_mkf=m.ms_mkf[order].astype(np.int32), ms_cam=ms_cam[order], ms_pt=m.ms_pt[order].astype(np.int32),
                   ms_uv=m.ms_uv[order].copy(), ms_level=level[order], true_world=m.pt_world.copy())
