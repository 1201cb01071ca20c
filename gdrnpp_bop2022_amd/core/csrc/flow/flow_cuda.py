"""Stand-in for the reference's compiled ``flow_cuda`` extension (core/csrc/flow/src/flow_cuda.cpp:30-47):
``forward(depth_src, depth_tgt, KT, Kinv) -> [flow, valid]`` on device tensors, served by gdrnpp_flow_forward."""
from .... import hip_lib


def forward(depth_src, depth_tgt, KT, Kinv):
    if not depth_src.is_cuda:
        raise RuntimeError("flow_cuda.forward: CPU tensors are not supported by this build (no CPU fallback)")
    flow, valid = hip_lib.flow_forward(depth_src.contiguous(), depth_tgt.contiguous(), KT.contiguous(), Kinv.contiguous())
    return [flow, valid]
