"""core/csrc/ransac_voting/ransac_voting_gpu.py:7-309 with the same signatures; the RANSAC rounds use the fused
vote+count kernel so the ``[round_hyp_num, vn, tn]`` u8 tensor (4.7 MB per round at 128x9x4096) is never
written; only the final single-hypothesis vote materialises flags (they feed the 2x2 least squares).
``idxs_fn`` injects the index draw for parity tests (the reference draws with ``Tensor.random_``).  ONE signature in this
module, always called with keywords: ``idxs_fn(bi=, round_idx=, round_hyp_num=, vn=, tn=) -> i32[round_hyp_num, vn, 2]``.
Like the reference, ``ransac_voting_layer[_v3]`` draw ONE index set per image and reuse it in every round (they pass
``round_idx=0``); ``estimate_voting_distribution_with_mean`` draws once per round."""
import numpy as np
import torch

from . import ransac_voting


def _foreground(cur_mask, vertex_bi, max_num):
    foreground_num = torch.sum(cur_mask)
    if foreground_num > max_num:
        selection = torch.zeros(cur_mask.shape, dtype=torch.float32, device=cur_mask.device).uniform_(0, 1)
        cur_mask = cur_mask & (selection < (max_num / foreground_num.float()))
    coords = torch.nonzero(cur_mask).float()[:, [1, 0]].contiguous()  # (x, y)
    vn = vertex_bi.shape[2]
    direct = vertex_bi.masked_select(cur_mask[:, :, None, None]).view(coords.shape[0], vn, 2).contiguous()
    return coords, direct


def _b_inv(b_mat):
    """ransac_voting_gpu.py:105-120: batched inverse by solving against the identity; a singular batch -> identity."""
    eye = b_mat.new_ones(b_mat.size(-1)).diag().expand_as(b_mat)
    try:
        return torch.linalg.solve(b_mat, eye)
    except Exception:  # noqa: BLE001  (https://github.com/zju3dv/clean-pvnet/issues/8)
        return eye


def _draw(idxs_fn, bi, round_hyp_num, vn, tn, device):
    """The hypothesis index draw — ONE set per image, made before the round loop and reused by every round
    (ransac_voting_gpu.py:48,164): the confidence test can only be met by the first round's hypotheses, later rounds
    re-vote the same set.  Reference behaviour, kept."""
    if idxs_fn is not None:
        return idxs_fn(bi=bi, round_idx=0, round_hyp_num=round_hyp_num, vn=vn, tn=tn)
    return torch.zeros([round_hyp_num, vn, 2], dtype=torch.int32, device=device).random_(0, tn)


def _ransac_rounds(mask, vertex, bi, round_hyp_num, inlier_thresh, confidence, max_iter, max_num, idxs_fn):
    """Rounds + inlier normal equations of one image -> (ATA [vn,2,2], ATb [vn,2])."""
    vn = vertex.shape[3]
    cur_mask = mask[bi].to(torch.bool)
    coords, direct = _foreground(cur_mask, vertex[bi], max_num)
    tn = coords.shape[0]
    idxs = _draw(idxs_fn, bi, round_hyp_num, vn, tn, mask.device)
    all_win_ratio = torch.zeros([vn], dtype=torch.float32, device=mask.device)
    all_win_pts = torch.zeros([vn, 2], dtype=torch.float32, device=mask.device)
    hyp_num, cur_iter = 0, 0
    while True:
        cur_hyp_pts = ransac_voting.generate_hypothesis(direct, coords, idxs)          # [hn,vn,2]
        cur_inlier_counts = ransac_voting.vote_count(direct, coords, cur_hyp_pts, inlier_thresh)  # [hn,vn]
        cur_win_idx = torch.argmax(cur_inlier_counts, 0)  # first maximum (torch.max leaves ties unspecified)
        cur_win_counts = cur_inlier_counts.gather(0, cur_win_idx[None])[0]
        cur_win_pts = cur_hyp_pts[cur_win_idx, torch.arange(vn, device=mask.device)]
        cur_win_ratio = cur_win_counts.float() / tn
        larger_mask = all_win_ratio < cur_win_ratio
        all_win_pts[larger_mask, :] = cur_win_pts[larger_mask, :]
        all_win_ratio[larger_mask] = cur_win_ratio[larger_mask]
        hyp_num += round_hyp_num
        cur_iter += 1
        cur_min_ratio = torch.min(all_win_ratio)
        if (1 - (1 - cur_min_ratio ** 2) ** hyp_num) > confidence or cur_iter > max_iter:
            break
    # mean intersection of the inliers of the winning hypothesis (:82-94 / :186-203)
    normal = torch.zeros_like(direct)
    normal[:, :, 0] = direct[:, :, 1]
    normal[:, :, 1] = -direct[:, :, 0]
    all_inlier = torch.zeros([1, vn, tn], dtype=torch.uint8, device=mask.device)
    ransac_voting.voting_for_hypothesis(direct, coords, all_win_pts[None].contiguous(), all_inlier, inlier_thresh)
    all_inlier = torch.squeeze(all_inlier.float(), 0)                 # [vn,tn]
    normal = normal.permute(1, 0, 2) * all_inlier[:, :, None]          # [vn,tn,2]
    bvec = torch.sum(normal * coords[None], 2)                          # [vn,tn]
    ATA = torch.matmul(normal.permute(0, 2, 1), normal)                 # [vn,2,2]
    ATb = torch.sum(normal * bvec[:, :, None], 1)                       # [vn,2]
    return ATA, ATb


def ransac_voting_layer(mask, vertex, round_hyp_num, inlier_thresh=0.999, confidence=0.99, max_iter=20, min_num=5,
                        max_num=30000, idxs_fn=None):
    """mask [b,h,w], vertex [b,h,w,vn,2] -> [b,vn,2]  (ransac_voting_gpu.py:7-104).  ``torch.inverse(ATA)``; an image
    whose inverse raises contributes zeros (:96-101)."""
    b, h, w, vn, _ = vertex.shape
    batch_win_pts = []
    for bi in range(b):
        if torch.sum(mask[bi].to(torch.bool)) < min_num:
            batch_win_pts.append(torch.zeros([1, vn, 2], dtype=torch.float32, device=mask.device))
            continue
        ATA, ATb = _ransac_rounds(mask, vertex, bi, round_hyp_num, inlier_thresh, confidence, max_iter, max_num, idxs_fn)
        try:
            win = torch.matmul(torch.inverse(ATA), ATb[:, :, None])         # [vn,2,1]
            batch_win_pts.append(win[None, :, :, 0])
        except Exception:  # noqa: BLE001  the reference's bare except
            batch_win_pts.append(torch.zeros([1, ATA.size(0), 2], device=ATA.device))
    return torch.cat(batch_win_pts)


def ransac_voting_layer_v3(mask, vertex, round_hyp_num, inlier_thresh=0.999, confidence=0.99, max_iter=20,
                           min_num=5, max_num=30000, idxs_fn=None):
    """ransac_voting_gpu.py:123-218 — as above, the final solve through ``b_inv`` (singular -> identity, :105-120)."""
    b, h, w, vn, _ = vertex.shape
    batch_win_pts = []
    for bi in range(b):
        if torch.sum(mask[bi].to(torch.bool)) < min_num:
            batch_win_pts.append(torch.zeros([1, vn, 2], dtype=torch.float32, device=mask.device))
            continue
        ATA, ATb = _ransac_rounds(mask, vertex, bi, round_hyp_num, inlier_thresh, confidence, max_iter, max_num, idxs_fn)
        win = torch.matmul(_b_inv(ATA), ATb[:, :, None])                    # [vn,2,1]
        batch_win_pts.append(win[None, :, :, 0])
    return torch.cat(batch_win_pts)


def estimate_voting_distribution_with_mean(mask, vertex, mean, round_hyp_num=256, min_hyp_num=4096, topk=128,
                                           inlier_thresh=0.99, min_num=5, max_num=30000, output_hyp=False,
                                           idxs_fn=None):
    """ransac_voting_gpu.py:221-309 -> (mean, cov[b,vn,2,2])."""
    b, h, w, vn, _ = vertex.shape
    all_hyp_pts, all_inlier_ratio = [], []
    for bi in range(b):
        cur_mask = mask[bi] == 1
        foreground = torch.sum(cur_mask)
        if foreground < min_num:
            all_hyp_pts.append(torch.zeros([1, min_hyp_num, vn, 2], dtype=torch.float32, device=mask.device))
            all_inlier_ratio.append(torch.ones([1, min_hyp_num, vn], dtype=torch.float32, device=mask.device))
            continue
        coords, direct = _foreground(cur_mask, vertex[bi], max_num)
        tn = coords.shape[0]
        cur_hyp_pts, cur_inlier_ratio = [], []
        for round_idx in range(int(np.ceil(min_hyp_num / round_hyp_num))):
            if idxs_fn is not None:
                idxs = idxs_fn(bi=bi, round_idx=round_idx, round_hyp_num=round_hyp_num, vn=vn, tn=tn)
            else:
                idxs = torch.zeros([round_hyp_num, vn, 2], dtype=torch.int32, device=mask.device).random_(0, tn)
            hyp_pts = ransac_voting.generate_hypothesis(direct, coords, idxs)
            counts = ransac_voting.vote_count(direct, coords, hyp_pts, inlier_thresh)
            cur_hyp_pts.append(hyp_pts)
            cur_inlier_ratio.append(counts.float() / float(tn))
        all_hyp_pts.append(torch.cat(cur_hyp_pts, 0)[None])
        all_inlier_ratio.append(torch.cat(cur_inlier_ratio, 0)[None])
    all_hyp_pts = torch.cat(all_hyp_pts, 0).permute(0, 2, 1, 3)          # b,vn,hn,2
    all_inlier_ratio = torch.cat(all_inlier_ratio, 0).permute(0, 2, 1).clone()  # b,vn,hn
    thresh = torch.max(all_inlier_ratio, 2)[0] - 0.1
    all_inlier_ratio[all_inlier_ratio < thresh[:, :, None]] = 0.0
    diff_pts = all_hyp_pts - mean[:, :, None]
    weighted_diff_pts = diff_pts * all_inlier_ratio[:, :, :, None]
    cov = torch.matmul(diff_pts.transpose(2, 3), weighted_diff_pts)
    cov = cov / (torch.sum(all_inlier_ratio, 2)[:, :, None, None] + 1e-3)
    return mean, cov
