"""Host-side TaylorCamera (Scaramuzza polynomial omni camera) constructor.

Mirrors /root/reference/src/TaylorCamera.cc:46-198 (ctor + RefreshParams) and :489-604
(FindInvPolyUsingRoots): the one-off fit of the inverse polynomial that the device code
then evaluates.  SURVEY.md section 2 row 2: "device Project/derivs; ctor stays host".
The fitted state is exported as the `mcp_camera` struct of include/mcp_ba.h.

Eigen::PolynomialSolver and TooN::SVD (TaylorCamera.cc:42,464-466,514-519) are replaced by
numpy.roots / numpy.linalg.lstsq; coefficients therefore agree with a real MCPTAM run only
to fitting precision, which is immaterial because every consumer (oracle and HIP path)
receives the same fitted numbers through the struct.
"""
import ctypes
import math

import numpy as np

MAX_INV_DEGREE = 30  # include/mcptam/TaylorCamera.h:74


class McpCamera(ctypes.Structure):
    """ctypes image of `mcp_camera` (include/mcp_ba.h)."""
    _fields_ = [
        ("params", ctypes.c_double * 9),
        ("image_size", ctypes.c_double * 2),
        ("affine", ctypes.c_double * 4),
        ("center", ctypes.c_double * 2),
        ("min_theta", ctypes.c_double),
        ("max_rho", ctypes.c_double),
        ("theta_mean", ctypes.c_double),
        ("theta_std", ctypes.c_double),
        ("n_inv", ctypes.c_int),
        ("pad_", ctypes.c_int),
        ("inv_coeffs", ctypes.c_double * 31),
    ]


def polyval_low_first(coeffs, x):
    """TaylorCamera::PolyVal (TaylorCamera.cc:472-486): coefficient of x^0 first."""
    val = np.zeros_like(np.asarray(x, dtype=np.float64))
    for c in coeffs[:0:-1]:
        val = (val + c) * x
    return val + coeffs[0]


class TaylorCamera:
    def __init__(self, params9, calib_size, fullscale_size, image_size, force_newton=False):
        # force_newton: behave as if no inverse polynomial of degree <= 30 met the 1e-4 limit, i.e. the reference's slow path
        # (TaylorCamera.cc:159-176): a linear inverse model as the starting point + Newton's method on the forward polynomial
        self.force_newton = bool(force_newton)
        self.params = np.asarray(params9, dtype=np.float64).copy()
        self.calib_size = np.asarray(calib_size, dtype=np.float64)
        self.fullscale_size = np.asarray(fullscale_size, dtype=np.float64)
        self.image_size = np.asarray(image_size, dtype=np.float64)
        self.refresh_params()

    # TaylorCamera::RefreshParams, TaylorCamera.cc:84-198
    def refresh_params(self):
        p = self.params
        self.poly = np.array([p[0], 0.0, p[1], p[2], p[3]])                     # :102-106
        self.poly_deriv_mod = self.poly * np.array([-1.0, 1.0, 1.0, 2.0, 3.0])  # :109-112
        scale = self.image_size / self.fullscale_size                           # :117-118
        fs_center = p[4:6] - (self.calib_size - self.fullscale_size) / 2        # :126-127
        self.center = fs_center * scale                                         # :132-133
        corner = np.maximum(fs_center, self.fullscale_size - fs_center - 1)     # :139-140
        self.largest_radius = math.sqrt(float(corner @ corner))                 # :146
        self.max_rho = 1.0 * self.largest_radius                                # :149
        self.min_theta = math.atan(float(polyval_low_first(self.poly, self.max_rho)) / self.max_rho)  # :154
        inv = None if self.force_newton else self.find_inv_poly_using_roots(-1, 1e-4)   # :159
        self.using_inverse_poly = inv is not None
        if inv is None:
            # :161-176: linear inverse model (degree-1 fit; it also refreshes theta mean / std) and the derivative of the
            # forward polynomial for Newton's method
            self.linear_inv_coeffs = self.find_inv_poly_using_roots(1, 0.1)
            self.poly_deriv = self.poly[1:] * np.arange(1, 5)
            inv = self.linear_inv_coeffs
        self.inv_coeffs = inv
        self.affine = np.array([[scale[0] * p[6], scale[1] * p[7]],
                                [scale[0] * p[8], scale[1] * 1.0]])             # :183-186
        self.affine_inv = np.linalg.inv(self.affine)

    # TaylorCamera::FindInvPolyUsingRoots, TaylorCamera.cc:489-604
    def find_inv_poly_using_roots(self, degree, err_limit):
        theta_start = -math.pi / 2 + 0.001
        theta_end = math.pi / 2 - 0.001
        step = 0.01
        n = int(math.ceil((theta_end - theta_start) / step)) + 1
        thetas = np.empty(n)
        thetas[0] = theta_start
        for i in range(1, n):
            thetas[i] = thetas[i - 1] + step
        rhos = np.full(n, -9999.0)
        for i in range(n):
            c = self.poly.copy()
            c[1] -= math.tan(thetas[i])
            hi = c[::-1]
            hi = hi[np.argmax(hi != 0):] if np.any(hi != 0) else hi   # numpy.roots wants a nonzero leading term
            roots = np.roots(hi)
            real = [r.real for r in roots if abs(r.imag) < 1e-12]     # Eigen realRoots threshold
            real = [r for r in real if not (r < 0.0 or r > self.max_rho)]
            if len(real) == 1:
                rhos[i] = real[0]
        keep = rhos != -9999.0
        th, rh = thetas[keep], rhos[keep]
        self.theta_mean = float(th.sum() / th.size)
        shifted = th - self.theta_mean
        self.theta_std = float(math.sqrt(float(shifted @ shifted) / shifted.size))
        ths = (th - self.theta_mean) / self.theta_std

        def polyfit(d):
            vand = np.vander(ths, d + 1, increasing=True)
            a, *_ = np.linalg.lstsq(vand, rh, rcond=None)
            return a

        if degree >= 0:
            return polyfit(degree)
        d = 2
        while d <= MAX_INV_DEGREE:
            a = polyfit(d)
            err = np.abs(rh - polyval_low_first(a, ths)).max()
            if err <= err_limit:
                return a
            d += 1
        return None

    # TaylorCamera::Project, TaylorCamera.cc:202-287 (vectorised; returns uv, invalid)
    def project(self, xc):
        xc = np.atleast_2d(np.asarray(xc, dtype=np.float64))
        norm = np.sqrt(xc[:, 0] ** 2 + xc[:, 1] ** 2)
        safe = np.where(norm == 0, 1.0, norm)
        theta = np.where(norm == 0, math.pi / 2, np.arctan(xc[:, 2] / safe))
        invalid = theta < self.min_theta
        rho = polyval_low_first(self.inv_coeffs, (theta - self.theta_mean) / self.theta_std)
        if not self.using_inverse_poly:
            rho = self._newton(xc[:, 2] / safe, rho)
        rho = np.where(norm == 0, 0.0, rho)
        cphi = np.where(norm == 0, 0.0, xc[:, 0] / safe)
        sphi = np.where(norm == 0, 0.0, xc[:, 1] / safe)
        d = np.stack([cphi * rho, sphi * rho], axis=1)
        uv = d @ self.affine.T + self.center
        invalid |= ~((uv[:, 0] >= 0) & (uv[:, 0] < self.image_size[0]) & (uv[:, 1] >= 0) & (uv[:, 1] < self.image_size[1]))
        return uv, invalid

    # TaylorCamera::FindRootWithNewton, TaylorCamera.cc:293-315 (error limit 0.01, at most 50 iterations, TaylorCamera.h:267)
    def _newton(self, tan_theta, rho0):
        rho = np.array(rho0, dtype=np.float64)
        active = np.ones(rho.shape, dtype=bool)
        for _ in range(50):
            c1 = self.poly[1] - tan_theta
            f = (((self.poly[4] * rho + self.poly[3]) * rho + self.poly[2]) * rho + c1) * rho + self.poly[0]
            d0 = self.poly_deriv[0] - tan_theta
            fp = ((self.poly_deriv[3] * rho + self.poly_deriv[2]) * rho + self.poly_deriv[1]) * rho + d0
            new = rho - f / fp
            err = np.abs(new - rho)
            rho = np.where(active, new, rho)
            active &= err > 0.01
            if not active.any():
                break
        return rho

    # TaylorCamera::UnProject, TaylorCamera.cc:319-347
    def unproject(self, uv):
        uv = np.atleast_2d(np.asarray(uv, dtype=np.float64))
        d = (uv - self.center) @ self.affine_inv.T
        rho = np.sqrt((d * d).sum(axis=1))
        v = np.stack([d[:, 0], d[:, 1], polyval_low_first(self.poly, rho)], axis=1)
        return v / np.linalg.norm(v, axis=1, keepdims=True)

    def to_struct(self):
        s = McpCamera()
        for i in range(9):
            s.params[i] = self.params[i]
        s.image_size[0], s.image_size[1] = self.image_size
        a = self.affine
        s.affine[0], s.affine[1], s.affine[2], s.affine[3] = a[0, 0], a[0, 1], a[1, 0], a[1, 1]
        s.center[0], s.center[1] = self.center
        s.min_theta, s.max_rho = self.min_theta, self.max_rho
        s.theta_mean, s.theta_std = self.theta_mean, self.theta_std
        if len(self.inv_coeffs) > 31:
            raise ValueError("inverse polynomial too long")
        # n_inv == 0 selects the Newton mode; inv_coeffs[0..1] then hold the linear inverse model (mv2LinearInvCoeffs)
        s.n_inv = len(self.inv_coeffs) if self.using_inverse_poly else 0
        for i, c in enumerate(self.inv_coeffs):
            s.inv_coeffs[i] = c
        return s


def camera_array(cams):
    arr = (McpCamera * len(cams))()
    for i, c in enumerate(cams):
        arr[i] = c.to_struct()
    return arr
