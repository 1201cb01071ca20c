"""core/csrc/torch_nndistance/torch_nndistance.py:13-88: ``nnd(xyz1, xyz2) -> (dist1, dist2)`` with autograd."""
import torch
from torch.autograd import Function

from . import torch_nndistance_aten as _C

__version__ = "1.0.0"


class NNDFunction(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        if not xyz1.is_cuda:
            raise RuntimeError("NNDFunction: CPU tensors are not supported by this build (no CPU fallback)")
        batchsize, n, _ = xyz1.size()
        _, m, _ = xyz2.size()
        xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
        dist1 = torch.zeros(batchsize, n, device=xyz1.device)
        dist2 = torch.zeros(batchsize, m, device=xyz1.device)
        idx1 = torch.zeros(batchsize, n, dtype=torch.int32, device=xyz1.device)
        idx2 = torch.zeros(batchsize, m, dtype=torch.int32, device=xyz1.device)
        _C.nnd_forward_cuda(xyz1, xyz2, dist1, dist2, idx1, idx2)
        ctx.save_for_backward(xyz1, xyz2, dist1, dist2, idx1, idx2)
        return dist1, dist2

    @staticmethod
    def backward(ctx, graddist1, graddist2):
        xyz1, xyz2, dist1, dist2, idx1, idx2 = ctx.saved_tensors
        gradxyz1 = torch.zeros_like(xyz1)
        gradxyz2 = torch.zeros_like(xyz2)
        _C.nnd_backward_cuda(xyz1, xyz2, gradxyz1, gradxyz2, graddist1.contiguous(), graddist2.contiguous(), idx1, idx2)
        return gradxyz1, gradxyz2


def nnd(xyz1, xyz2):
    return NNDFunction.apply(xyz1, xyz2)
