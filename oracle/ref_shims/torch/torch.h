// ORACLE build shim — a minimal stand-in for <torch/torch.h> so that the reference's
// core/csrc/torch_nndistance/src/nnd_cpu.cpp compiles UNMODIFIED without libtorch
// (its algorithm, nnsearch(), is plain C++; only the tensor accessors are needed).
#pragma once
#include <cstdint>
namespace at {
struct Tensor {
  void* ptr;
  long sizes[4];
  long size(int i) const { return sizes[i]; }
  template <typename T> T* data_ptr() const { return static_cast<T*>(ptr); }
};
}  // namespace at
struct OracleDummyModule {
  template <typename F> void def(const char*, F, const char*) {}
};
#define TORCH_EXTENSION_NAME oracle_nnd_ref
#define PYBIND11_MODULE(name, m) static void oracle_pybind_stub_##name(OracleDummyModule& m)
