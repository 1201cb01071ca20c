"""Drop-in for the torch extension module ``ransac_voting`` (core/csrc/ransac_voting/src/ransac_voting.cpp:112-117):
the same four functions on torch tensors, plus the fused ``vote_count``.  Input checks mirror ``CHECK_INPUT``
(:7-19: device + contiguous -> RuntimeError); unlike the reference, launches go to torch's CURRENT stream."""
import torch

from .... import hip_lib


def _chk(x, dtype, name):
    return hip_lib._dev(x, dtype, name)


def _dims(direct, coords, third, last):
    tn, vn, two = direct.shape
    assert two == 2 and coords.shape == (tn, 2) and third.shape[1] == vn and third.shape[2] == last
    return tn, vn, third.shape[0]


def generate_hypothesis(direct, coords, idxs):
    tn, vn, hn = _dims(direct, coords, idxs, 2)
    hypo_pts = torch.empty((hn, vn, 2), dtype=torch.float32, device=direct.device)
    hip_lib._check(hip_lib.load().gdrnpp_generate_hypothesis(
        _chk(direct, torch.float32, "direct"), _chk(coords, torch.float32, "coords"), _chk(idxs, torch.int32, "idxs"),
        hypo_pts.data_ptr(), tn, vn, hn, hip_lib._stream()), "generate_hypothesis")
    return hypo_pts


def voting_for_hypothesis(direct, coords, hypo_pts, inliers, inlier_thresh):
    tn, vn, hn = _dims(direct, coords, hypo_pts, 2)
    assert inliers.shape == (hn, vn, tn)
    hip_lib._check(hip_lib.load().gdrnpp_voting_for_hypothesis(
        _chk(direct, torch.float32, "direct"), _chk(coords, torch.float32, "coords"),
        _chk(hypo_pts, torch.float32, "hypo_pts"), _chk(inliers, torch.uint8, "inliers"), tn, vn, hn,
        float(inlier_thresh), hip_lib._stream()), "voting_for_hypothesis")


def generate_hypothesis_vanishing_point(direct, coords, idxs):
    tn, vn, hn = _dims(direct, coords, idxs, 2)
    hypo_pts = torch.empty((hn, vn, 3), dtype=torch.float32, device=direct.device)
    hip_lib._check(hip_lib.load().gdrnpp_generate_hypothesis_vanishing_point(
        _chk(direct, torch.float32, "direct"), _chk(coords, torch.float32, "coords"), _chk(idxs, torch.int32, "idxs"),
        hypo_pts.data_ptr(), tn, vn, hn, hip_lib._stream()), "generate_hypothesis_vanishing_point")
    return hypo_pts


def voting_for_hypothesis_vanishing_point(direct, coords, hypo_pts, inliers, inlier_thresh):
    tn, vn, hn = _dims(direct, coords, hypo_pts, 3)
    assert inliers.shape == (hn, vn, tn)
    hip_lib._check(hip_lib.load().gdrnpp_voting_for_hypothesis_vanishing_point(
        _chk(direct, torch.float32, "direct"), _chk(coords, torch.float32, "coords"),
        _chk(hypo_pts, torch.float32, "hypo_pts"), _chk(inliers, torch.uint8, "inliers"), tn, vn, hn,
        float(inlier_thresh), hip_lib._stream()), "voting_for_hypothesis_vanishing_point")


def vote_count(direct, coords, hypo_pts, inlier_thresh):
    """counts i32[hn,vn] == voting_for_hypothesis(...) summed over pixels, without the [hn,vn,tn] tensor."""
    homo = hypo_pts.shape[2] == 3
    tn, vn, hn = _dims(direct, coords, hypo_pts, 3 if homo else 2)
    counts = torch.empty((hn, vn), dtype=torch.int32, device=direct.device)
    hip_lib._check(hip_lib.load().gdrnpp_vote_count(
        _chk(direct, torch.float32, "direct"), _chk(coords, torch.float32, "coords"),
        _chk(hypo_pts, torch.float32, "hypo_pts"), counts.data_ptr(), tn, vn, hn, float(inlier_thresh),
        1 if homo else 0, hip_lib._stream()), "vote_count")
    return counts
