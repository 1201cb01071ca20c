"""The reference-named Python surface (core.csrc.* shims) on the GPU against the oracle (-m gpu)."""
import numpy as np
import pytest
import torch

from oracle import postproc as P

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_fps_utils_like_reference_call_site():
    """core/utils/data_utils.py:255-264 / gdrn_evaluator.py:105-113 style use."""
    from gdrnpp_bop2022_amd.core.csrc.fps.fps_utils import farthest_point_sampling

    rng = np.random.default_rng(0)
    pts = (rng.standard_normal((4000, 3)) * 0.05).astype(np.float32)
    out = farthest_point_sampling(pts, 8, init_center=True)
    assert out.shape == (8, 3) and np.array_equal(out, pts[P.fps(pts, 8, True)])
    out = farthest_point_sampling(pts, 8, init_center=False)
    idx0 = int(np.flatnonzero((pts == out[0]).all(1))[0])
    assert np.array_equal(out, pts[P.fps(pts, 8, False, idx0)])


def test_un_pnp_utils_like_pose_from_upnp():
    """gdrn_evaluator.py:612-628 (pose_from_upnp): caller-supplied initialiser and the default EPnP one."""
    from gdrnpp_bop2022_amd.core.csrc.uncertainty_pnp.un_pnp_utils import uncertainty_pnp, uncertainty_pnp_v2

    rng = np.random.default_rng(1)
    K = np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1.0]])
    rt = np.array([0.4, -0.3, 0.2, 0.03, -0.02, 0.9])
    p3 = rng.uniform(-0.05, 0.05, (9, 3))
    th = np.linalg.norm(rt[:3]); k = rt[:3] / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    X = p3 @ R.T + rt[3:]
    p2 = np.stack([K[0, 0] * X[:, 0] / X[:, 2] + K[0, 2], K[1, 1] * X[:, 1] / X[:, 2] + K[1, 2]], 1)
    p2 = p2 + rng.normal(0, 0.5, p2.shape)
    cov = np.stack([np.diag(rng.uniform(0.5, 4, 2)) for _ in range(9)])
    w = np.stack([[1 / np.sqrt(c[0, 0]), 0.0, 1 / np.sqrt(c[1, 1])] for c in cov])
    init = rt + rng.uniform(0, 0.05, 6)
    Rt = uncertainty_pnp(p2, w, p3, K, init_rt=init)
    o = P.uncertainty_pnp(p2, p3, w, K, init)
    assert Rt.shape == (3, 4)
    np.testing.assert_allclose(Rt[:, 3], o[3:], atol=1e-9)
    th2 = np.linalg.norm(o[:3]); k2 = o[:3] / th2
    K2 = np.array([[0, -k2[2], k2[1]], [k2[2], 0, -k2[0]], [-k2[1], k2[0], 0]])
    np.testing.assert_allclose(Rt[:, :3], np.eye(3) + np.sin(th2) * K2 + (1 - np.cos(th2)) * K2 @ K2, atol=1e-9)
    np.testing.assert_allclose(Rt[:, :3], R, atol=5e-2)
    Rt2 = uncertainty_pnp_v2(p2, cov, p3, K, init_rt=init)
    assert np.isfinite(Rt2).all()
    # reference call signature (no init_rt): EPnP on the four best-weighted points seeds the LM (un_pnp_utils.py:27-44)
    Rt3 = uncertainty_pnp(p2, w, p3, K)
    np.testing.assert_allclose(Rt3[:, :3], R, atol=5e-2)
    np.testing.assert_allclose(Rt3[:, 3], rt[3:], atol=3e-2)
    Rt4 = uncertainty_pnp_v2(p2, cov, p3, K)
    np.testing.assert_allclose(Rt4[:, 3], rt[3:], atol=3e-2)


def test_ransac_voting_layer_replays_reference_draw(golden_dir):
    """The shim against the REFERENCE's ransac_voting_layer (executed from source over its host-compiled kernels,
    tests/golden/make_golden_pyref.py): the recorded single index draw per image is replayed through idxs_fn and the
    voted keypoints must agree; an image below min_num foreground pixels gives zeros."""
    from gdrnpp_bop2022_amd.core.csrc.ransac_voting.ransac_voting_gpu import ransac_voting_layer, ransac_voting_layer_v3

    g = np.load(f"{golden_dir}/pyref_golden.npz")
    mask, vertex, win_ref, draws = g["rv_mask"], g["rv_vertex"], g["rv_win"], g["rv_idxs"]
    calls = []

    def idxs_fn(bi, round_idx, round_hyp_num, vn, tn):
        calls.append(bi)
        assert round_idx == 0 and draws[bi].shape == (round_hyp_num, vn, 2) and draws[bi].max() < tn
        return torch.from_numpy(draws[bi]).to(DEV)

    for layer in (ransac_voting_layer, ransac_voting_layer_v3):
        calls.clear()
        out = layer(torch.from_numpy(mask).to(DEV), torch.from_numpy(vertex).to(DEV), 128, inlier_thresh=0.99, max_iter=5,
                    idxs_fn=idxs_fn).cpu().numpy()
        assert calls == [0, 1]                               # ONE draw per image with enough foreground, none for image 2
        np.testing.assert_allclose(out[:2], win_ref[:2], atol=2e-3)   # 2x2 inverse in fp32, different summation order
        assert np.array_equal(out[2], np.zeros_like(out[2]))


def test_ransac_voting_layer_matches_oracle():
    from gdrnpp_bop2022_amd.core.csrc.ransac_voting.ransac_voting_gpu import (
        estimate_voting_distribution_with_mean, ransac_voting_layer_v3)

    rng = np.random.default_rng(2)
    b, h, w, vn, hn = 2, 64, 64, 9, 128
    mask = np.zeros((b, h, w), np.float32)
    mask[:, 16:48, 12:52] = 1
    mask[1, :, :] = 0
    mask[1, 10, 10:13] = 1  # < min_num foreground -> zeros
    kp = rng.uniform(5, 60, (b, vn, 2)).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    pix = np.stack([xx, yy], -1).astype(np.float32)
    vertex = kp[:, None, None] - pix[None, :, :, None]
    vertex = vertex / np.maximum(np.linalg.norm(vertex, axis=-1, keepdims=True), 1e-6)
    vertex = (vertex + rng.normal(0, 0.02, vertex.shape)).astype(np.float32)
    tn0 = int(mask[0].sum())
    draw = rng.integers(0, tn0, (hn, vn, 2)).astype(np.int32)

    def idxs_fn(bi, round_idx, round_hyp_num, vn, tn):
        return torch.from_numpy(draw).to(DEV)

    out = ransac_voting_layer_v3(torch.from_numpy(mask).to(DEV), torch.from_numpy(vertex).to(DEV), hn, idxs_fn=idxs_fn)
    out = out.cpu().numpy()
    ref, iters = P.ransac_voting_layer(mask[0], vertex[0], [draw] * 25)   # the same set in every round, like the reference
    np.testing.assert_allclose(out[0], ref, atol=2e-3)      # fp32 torch LSQ vs fp64 NumPy
    assert np.abs(out[0] - kp[0]).max() < 1.0                # the keypoints are recovered
    assert np.array_equal(out[1], np.zeros((vn, 2), np.float32))
    rounds = []

    def idxs_fn_rounds(bi, round_idx, round_hyp_num, vn, tn):    # same keyword signature, one draw per round here
        rounds.append((bi, round_idx))
        return torch.from_numpy(np.random.default_rng(round_idx).integers(0, tn, (round_hyp_num, vn, 2)).astype(np.int32)).to(DEV)

    mean, cov = estimate_voting_distribution_with_mean(torch.from_numpy(mask).to(DEV), torch.from_numpy(vertex).to(DEV),
                                                       torch.from_numpy(out).to(DEV), round_hyp_num=hn, min_hyp_num=512,
                                                       idxs_fn=idxs_fn_rounds)
    assert rounds == [(0, r) for r in range(4)]
    cov = cov.cpu().numpy()
    assert cov.shape == (b, vn, 2, 2) and np.isfinite(cov).all()
    assert (np.linalg.eigvalsh(cov[0]) > -1e-4).all()


def test_nnd_autograd_wrapper():
    from gdrnpp_bop2022_amd.core.csrc.torch_nndistance.torch_nndistance import nnd

    torch.manual_seed(0)
    x1 = torch.rand(4, 300, 3, device=DEV, requires_grad=True)
    x2 = torch.rand(4, 500, 3, device=DEV, requires_grad=True)
    d1, d2 = nnd(x1, x2)
    (d1.mean() + d2.mean()).backward()
    # plain PyTorch fp32 reference of the same op
    y1 = x1.detach().clone().requires_grad_(True)
    y2 = x2.detach().clone().requires_grad_(True)
    D = ((y1[:, :, None] - y2[:, None]) ** 2).sum(-1)
    r1, r2 = D.min(2)[0], D.min(1)[0]
    (r1.mean() + r2.mean()).backward()
    torch.testing.assert_close(d1, r1, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(d2, r2, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(x1.grad, y1.grad, rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(x2.grad, y2.grad, rtol=1e-4, atol=1e-7)


def test_egl_renderer_shim_pc_obj_and_pc_cam():
    """EGLRenderer.render(obj_ids, poses, K, pc_obj_tensor=…, pc_cam_tensor=…) as engine_utils.py:131-172 calls it:
    object-space xyz / camera-space xyz + depth, two objects composited by nearest depth."""
    from gdrnpp_bop2022_amd import synthetic as S
    from gdrnpp_bop2022_amd.lib.egl_renderer.egl_renderer_v3 import EGLRenderer

    rng = np.random.default_rng(5)
    verts, faces, ext = S.make_models(2, rng, subdiv=3)
    ren = EGLRenderer(models=[{"pts": verts[i], "faces": faces[i]} for i in range(2)], height=64, width=64,
                      znear=0.1, zfar=10.0)
    K = np.array([[300.0, 0, 32], [0, 300.0, 32], [0, 0, 1]], np.float32)
    poses = [np.hstack([S.random_rotation(rng), [[0.01], [0.0], [0.7]]]).astype(np.float32),
             np.hstack([S.random_rotation(rng), [[-0.03], [0.02], [0.9]]]).astype(np.float32)]
    pc_obj = torch.zeros(64, 64, 4, device=DEV)
    pc_cam = torch.zeros(64, 64, 4, device=DEV)
    ren.render([0, 1], poses, K=K, pc_obj_tensor=pc_obj, pc_cam_tensor=pc_cam)
    d = [P.render_depth(verts[i], faces[i], K, poses[i][:, :3], poses[i][:, 3].astype(np.float64), 64, z_near=0.1,
                        z_far=10.0, want_xyz=True) for i in range(2)]
    d0 = np.where(d[0][0] > 0, d[0][0], np.inf); d1 = np.where(d[1][0] > 0, d[1][0], np.inf)
    z = np.minimum(d0, d1); z = np.where(np.isfinite(z), z, 0).astype(np.float32)
    assert np.array_equal(pc_cam[:, :, 2].cpu().numpy(), z)
    xyz = np.where((d0 <= d1)[..., None], d[0][1], d[1][1]) * (z > 0)[..., None]
    np.testing.assert_allclose(pc_obj[:, :, :3].cpu().numpy(), xyz, atol=1e-6)
    assert np.array_equal(pc_obj[:, :, 3].cpu().numpy(), (z > 0).astype(np.float32))
    # camera-space points back-project to the pixel centres
    m = z > 0
    u = pc_cam[:, :, 0].cpu().numpy()[m] / z[m] * 300 + 32
    jj, ii = np.mgrid[0:64, 0:64]
    np.testing.assert_allclose(u, ii[m] + 0.5, atol=1e-3)


def test_cpp_egl_renderer_class_boundary():
    """CppEGLRenderer(w, h, dev).init / query / map_tensor(tex_id, w, h, dev_ptr) / draw / release
    (lib/egl_renderer/cpp/egl_renderer.cpp:99-311): map_tensor copies an attachment in GL row order into a raw device
    pointer; flipping the rows (what egl_renderer_v3.py does after every map_tensor) restores the image."""
    from gdrnpp_bop2022_amd.lib.egl_renderer import CppEGLRenderer

    r = CppEGLRenderer.CppEGLRenderer(64, 48, 0)
    with pytest.raises(RuntimeError, match="init"):
        r.write_attachment(4, torch.zeros(48, 64, 4, device=DEV))
    assert r.init() == 0
    img = torch.rand(48, 64, 4, device=DEV)
    r.write_attachment(4, img)
    out = torch.zeros(48, 64, 4, device=DEV)
    r.map_tensor(4, 64, 48, out.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(out, torch.flip(img, (0,))) and torch.equal(torch.flip(out, (0,)), img)
    with pytest.raises(RuntimeError, match="nothing was rendered"):
        r.map_tensor(5, 64, 48, out.data_ptr())
    with pytest.raises(RuntimeError, match="size"):
        r.map_tensor(4, 32, 48, out.data_ptr())
    a = np.zeros((2, 3), np.float32)
    r.draw(a)
    assert (a == 42).all()
    r.query()
    r.release()
    with pytest.raises(RuntimeError):
        r.map_tensor(4, 64, 48, out.data_ptr())
