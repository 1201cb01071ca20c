"""gdrnpp_bop2022_amd — MI355X-native GDRNPP inference hot path.

Layout (only what the path needs):
  csrc/            hand-written HIP kernels for gfx950 + the flat C ABI (include/gdrnpp_hip.h)
  hip_lib.py       ctypes binding of libgdrnpp_hip.so (fails loudly when the library is missing)
  core/csrc/...    import-compatible shims of the reference's op modules (fps_utils, un_pnp_utils,
                   ransac_voting_gpu, torch_nndistance) routed through the C ABI
  gdrn_modeling/   GDRN_Net forward (PyTorch-ROCm) + the device-resident evaluator / engine
  synthetic.py     seeded synthetic workload
"""
__version__ = "0.1.0"
