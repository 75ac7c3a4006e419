"""Synthetic frames and map points for the KeyFrame / Tracker image path:
band-limited noise + random rectangles on a plane, rendered through the TaylorCamera at two
nearby poses so that patches of the first view are findable in the second."""
import math

import numpy as np
from scipy import ndimage

from .synth import DEFAULT_CAM_PARAMS, DEFAULT_SEED, so3_exp
from .taylor_camera import TaylorCamera


def make_texture(seed=DEFAULT_SEED, n=1536):
    rng = np.random.default_rng([seed, 77])
    t = ndimage.gaussian_filter(rng.normal(size=(n, n)), 2.5)
    t = (t - t.min()) / (t.max() - t.min()) * 150 + 40
    for _ in range(260):
        x0, y0 = rng.integers(0, n - 40, 2)
        ww, hh = rng.integers(8, 90, 2)
        t[y0:y0 + hh, x0:x0 + ww] = rng.uniform(10, 245)
    t = ndimage.gaussian_filter(t, 0.8)
    return np.clip(t, 0, 255)


def render_plane(cam, R_cw, t_cw, tex, depth=6.0, tex_scale=70.0):
    """Image of the textured plane z_w = depth seen from the camera pose (R_cw, t_cw)."""
    w, h = int(cam.image_size[0]), int(cam.image_size[1])
    uu, vv = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    rays_c = cam.unproject(np.stack([uu.ravel(), vv.ravel()], axis=1))
    rays_w = rays_c @ R_cw            # R_cw^T applied to each ray
    origin = -R_cw.T @ t_cw
    s = (depth - origin[2]) / np.where(np.abs(rays_w[:, 2]) < 1e-9, 1e-9, rays_w[:, 2])
    X = origin + rays_w * s[:, None]
    n = tex.shape[0]
    tx = X[:, 0] * tex_scale + n / 2
    ty = X[:, 1] * tex_scale + n / 2
    img = ndimage.map_coordinates(tex, [ty, tx], order=1, mode="reflect")
    img[s <= 0] = 0
    return np.clip(np.rint(img), 0, 255).astype(np.uint8).reshape(h, w)


def make_tracking_scene(seed=DEFAULT_SEED, size=(640, 480), depth=6.0):
    cam = TaylorCamera(DEFAULT_CAM_PARAMS[:4] + (size[0] / 2.0, size[1] / 2.0) + DEFAULT_CAM_PARAMS[6:], size, size, size)
    tex = make_texture(seed)
    RA, tA = np.eye(3), np.zeros(3)
    RB = so3_exp(np.array([0.004, -0.006, 0.01]))
    tB = np.array([0.12, -0.05, 0.08])
    imgA = render_plane(cam, RA, tA, tex, depth)
    imgB = render_plane(cam, RB, tB, tex, depth)
    return dict(cam=cam, imgA=imgA, imgB=imgB, poseA=(RA, tA), poseB=(RB, tB), depth=depth, tex=tex)


def make_map_points(cam, kf, kf_oracle, pose, depth, per_level=(400, 300, 200, 100), normal_c=(0.0, 0.0, -1.0)):
    """Map points at the FAST candidates of keyframe `kf` (pose = CamFromWorld of that keyframe),
    with the patch vectors of MapPoint::RefreshPixelVectors (/root/reference/src/MapPoint.cc:62-87)."""
    R, t = pose
    nrm = np.asarray(normal_c)
    pts = []
    for level, cap in enumerate(per_level):
        pos, _ = kf.Candidates(level)
        sc = 1 << level
        for c in pos[:cap]:
            cen = (np.array(c, dtype=np.float64) + 0.5) * sc - 0.5                   # LevelZeroPos
            right = (np.array([c[0] + 1, c[1]], dtype=np.float64) + 0.5) * sc - 0.5
            down = (np.array([c[0], c[1] + 1], dtype=np.float64) + 0.5) * sc - 0.5
            rc, rr, rd = cam.unproject(np.stack([cen, right, down]))
            Xc = rc * (depth - 0.0) / rc[2] if np.allclose(R, np.eye(3)) and np.allclose(t, 0) else None
            if Xc is None:
                origin = -R.T @ t
                rw = R.T @ rc
                Xw = origin + rw * (depth - origin[2]) / rw[2]
                Xc = R @ Xw + t
            Xw = R.T @ (Xc - t)
            cam_height = abs(Xc @ nrm)
            cen_p = rc * cam_height / abs(rc @ nrm)
            right_p = rr * cam_height / abs(rr @ nrm)
            down_p = rd * cam_height / abs(rd @ nrm)
            pts.append(dict(world_pos=Xw, pixel_right_w=R.T @ (right_p - cen_p), pixel_down_w=R.T @ (down_p - cen_p),
                            source_kf=kf, source_kf_oracle=kf_oracle, source_level=level, center=(int(c[0]), int(c[1])), fixed=0))
    return pts


def hypothesis_point(cam, kf, kf_oracle, pose, center, level, scale_along_ray, normal_c=(0.0, 0.0, -1.0)):
    """A hypothesised map point on the view ray of candidate `center` (level coordinates) of keyframe `kf`, `scale_along_ray`
    metres from the camera -- what MapMakerServerBase::AddPointEpipolar builds for every step along the epipolar arc
    (/root/reference/src/MapMakerServerBase.cc:724-760) before it hands the point to PatchFinder."""
    R, t = pose
    nrm = np.asarray(normal_c)
    sc = 1 << level
    c = np.asarray(center, dtype=np.float64)
    cen = (c + 0.5) * sc - 0.5
    right = cen + np.array([sc, 0.0])
    down = cen + np.array([0.0, sc])
    rc, rr, rd = cam.unproject(np.stack([cen, right, down]))
    Xc = rc * scale_along_ray
    Xw = R.T @ (Xc - t)
    cam_height = abs(Xc @ nrm)
    cen_p = rc * cam_height / abs(rc @ nrm)
    right_p = rr * cam_height / abs(rr @ nrm)
    down_p = rd * cam_height / abs(rd @ nrm)
    return dict(world_pos=Xw, pixel_right_w=R.T @ (right_p - cen_p), pixel_down_w=R.T @ (down_p - cen_p),
                source_kf=kf, source_kf_oracle=kf_oracle, source_level=level, center=(int(center[0]), int(center[1])), fixed=0)


def make_smooth_scene(seed=5, size=(640, 480)):
    """A frame with large-scale structure (what a SmallBlurryImage can see after 16:1 reduction): low-pass noise plus two
    rectangles.  Returns (image, image rotated by 3 degrees and shifted, an unrelated frame)."""
    w, h = size
    out = []
    for k in range(2):
        rng = np.random.default_rng([seed, k])
        base = ndimage.gaussian_filter(rng.normal(size=(h, w)), 30)
        base = (base - base.min()) / (base.max() - base.min()) * 200 + 20
        base[h // 5:h // 2, w // 4:w // 2] += 30
        base[5 * h // 8:7 * h // 8, 5 * w // 8:7 * w // 8] -= 25
        out.append(np.clip(base, 0, 255).astype(np.uint8))
    rot = ndimage.rotate(out[0].astype(float), 3.0, reshape=False, order=1, mode="nearest")
    rot = ndimage.shift(rot, (6, -9), order=1, mode="nearest").astype(np.uint8)
    return out[0], rot, out[1]
