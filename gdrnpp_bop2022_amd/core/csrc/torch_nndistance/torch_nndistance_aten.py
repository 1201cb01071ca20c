"""Drop-in for the torch extension module ``torch_nndistance_aten`` (nnd_cuda.cpp:86-89): caller-allocated outputs,
returns 1 on success (the reference returns 0 after printing a CUDA error; here a RuntimeError is raised)."""
from .... import hip_lib


def nnd_forward_cuda(xyz1, xyz2, dist1, dist2, idx1, idx2):
    return hip_lib.nnd_forward(xyz1, xyz2, dist1, dist2, idx1, idx2)


def nnd_backward_cuda(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
    return hip_lib.nnd_backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2)
