"""Closed-form checks that pin the software-rasteriser oracle's geometry (the GL renderer it
stands in for cannot run here): plane depth, sphere depth, half-pixel sampling, near/far, both faces."""
import numpy as np

from gdrnpp_bop2022_amd import synthetic as S
from oracle import postproc as P

K = np.array([[100.0, 0, 32.0], [0, 100.0, 32.0], [0, 0, 1]], np.float32)
I3 = np.eye(3, dtype=np.float32)


def _quad(z=0.0, half=10.0):
    v = np.array([[-half, -half, z], [half, -half, z], [half, half, z], [-half, half, z]], np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    return v, f


def test_fronto_parallel_plane_has_constant_depth():
    v, f = _quad()
    d = P.render_depth(v, f, K, I3, [0, 0, 2.0], 64)
    assert np.all(d == np.float32(2.0))


def test_tilted_plane_matches_ray_plane_intersection():
    v, f = _quad()
    ang = 0.5
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
    t = np.array([0.1, -0.05, 3.0])
    d = P.render_depth(v, f, K, R, t, 64)
    n = R.astype(np.float64)[:, 2]           # plane normal in camera space
    jj, ii = np.mgrid[0:64, 0:64]
    ray = np.stack([(ii + 0.5 - 32.0) / 100.0, (jj + 0.5 - 32.0) / 100.0, np.ones_like(ii, float)], -1)
    z = (n @ t) / (ray @ n)                  # pixel (row j, col i) sampled at (i+0.5, j+0.5)
    np.testing.assert_allclose(d, z.astype(np.float32), rtol=2e-7)


def test_sphere_depth_close_to_analytic():
    sv, sf = S.icosphere(4)
    r = 0.1
    v = (sv * r).astype(np.float32)
    t = np.array([0.0, 0.0, 1.0])
    d = P.render_depth(v, sf, K, I3, t, 64)
    jj, ii = np.mgrid[0:64, 0:64]
    ray = np.stack([(ii + 0.5 - 32.0) / 100.0, (jj + 0.5 - 32.0) / 100.0, np.ones_like(ii, float)], -1)
    a = (ray * ray).sum(-1)
    b = -2 * ray[..., 2] * t[2]
    disc = b * b - 4 * a * (t[2] ** 2 - r * r)
    hit = disc > 0
    z = np.where(hit, (-b - np.sqrt(np.abs(disc))) / (2 * a), 0.0)
    inner = disc > 0.01  # away from the silhouette: facet sagitta (~5e-5 m) / cos(incidence)
    assert inner.sum() > 200
    assert np.all(d[inner] > 0)
    assert np.abs(d[inner] - z[inner]).max() < 2.5e-4
    assert np.all(d[inner] >= z[inner].astype(np.float32))  # inscribed polyhedron lies behind the sphere
    assert np.all(d[~hit] == 0)   # an inscribed polyhedron never covers pixels outside the true silhouette


def test_near_far_clipping_and_back_faces():
    v, f = _quad()
    assert np.all(P.render_depth(v, f, K, I3, [0, 0, 0.05], 64) == 0)      # nearer than 0.1
    assert np.all(P.render_depth(v, f, K, I3, [0, 0, 150.0], 64) == 0)     # beyond 100
    flip = np.diag([1.0, -1.0, -1.0]).astype(np.float32)                    # shows the back face
    assert np.all(P.render_depth(v, f, K, flip, [0, 0, 2.0], 64) == np.float32(2.0))


def test_object_space_xyz_is_inverse_of_pose():
    sv, sf = S.icosphere(3)
    v = (sv * np.array([0.05, 0.08, 0.1])).astype(np.float32)
    R = S.random_rotation(np.random.default_rng(0)).astype(np.float32)
    t = np.array([0.02, -0.01, 0.6])
    d, xyz = P.render_depth(v, sf, K * np.array([[5], [5], [1]], np.float32) - np.array(
        [[0, 0, 128], [0, 0, 128], [0, 0, 0]], np.float32), R, t, 64, want_xyz=True)
    Kc = K * np.array([[5], [5], [1]], np.float32) - np.array([[0, 0, 128], [0, 0, 128], [0, 0, 0]], np.float32)
    jj, ii = np.mgrid[0:64, 0:64]
    m = d > 0
    assert m.sum() > 100
    cam = np.stack([(ii + 0.5 - Kc[0, 2]) / Kc[0, 0] * d, (jj + 0.5 - Kc[1, 2]) / Kc[1, 1] * d, d], -1)
    obj = (cam - t) @ R.astype(np.float64)   # R^T (X - t)
    assert np.abs(obj[m] - xyz[m]).max() < 1e-5
