"""core/csrc/flow/flow_torch.py:15-43: ``flow(depth_src, depth_tgt, pose_src, pose_tgt, K) -> (flow, valid)``.

The relative motion se3_src2tgt = pose_tgt * pose_src^-1 (core/utils/pose_utils.py:771-803) and KT = K @ se3 are
small batched 3x4 products done with torch on the device; the per-pixel work is the HIP kernel."""
import torch
from torch.autograd import Function

from . import flow_cuda


def _se3_src2tgt(pose_src, pose_tgt):
    r_inv = pose_src[:, :3, :3].transpose(1, 2)
    t_inv = -torch.matmul(r_inv, pose_src[:, :3, 3:4])
    r_new = torch.matmul(pose_tgt[:, :3, :3], r_inv)
    t_new = torch.matmul(pose_tgt[:, :3, :3], t_inv) + pose_tgt[:, :3, 3:4]
    return torch.cat([r_new, t_new], dim=-1)


class FlowFunction(Function):
    @staticmethod
    def forward(ctx, depth_src, depth_tgt, pose_src, pose_tgt, K):
        """depth_*: Bx1xHxW, pose_*: Bx3x4, K: Bx3x3 -> flow Bx2xHxW (dh, dw), valid Bx1xHxW."""
        KT = torch.matmul(K, _se3_src2tgt(pose_src, pose_tgt))
        Kinv = K.inverse().contiguous()
        flow, valid = flow_cuda.forward(depth_src, depth_tgt, KT, Kinv)
        return flow, valid


flow = FlowFunction.apply
