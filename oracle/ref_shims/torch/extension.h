// ORACLE build shim — a minimal stand-in for <torch/extension.h> so that the reference's
// core/csrc/flow/src/flow_cpu.cpp compiles UNMODIFIED without libtorch.  Only its template
// flow_kernel<scalar_t>() (plain C++) is called; the tensor wrapper around it just has to parse.
// The real header reaches <math.h> and <stdlib.h> (through pybind11 -> Python.h); libstdc++'s C++ wrappers of those bring
// the float overloads of abs()/round() into the global namespace.  They are included here the same way: with <cmath>
// alone the reference's unqualified `abs(z_proj - d_tgt)` would bind to int abs(int) and accept any |dz| < 1 m.
#pragma once
#include <math.h>
#include <stdlib.h>
#include <cmath>
#include <cstdlib>
#include <initializer_list>
#include <vector>
namespace torch {
enum class ScalarType { Double, Float };
struct TensorOptions {};
struct Tensor {
  void* ptr = nullptr;
  long sizes_[4] = {0, 0, 0, 0};
  ScalarType st = ScalarType::Float;
  long size(int i) const { return sizes_[i]; }
  TensorOptions options() const { return {}; }
  ScalarType scalar_type() const { return st; }
  template <typename T> T* data() const { return static_cast<T*>(ptr); }
};
inline Tensor zeros(std::initializer_list<long>, TensorOptions) { return Tensor{}; }
}  // namespace torch
struct OracleFlowDummyModule {
  template <typename F> void def(const char*, F, const char*) {}
};
#define TORCH_EXTENSION_NAME oracle_flow_ref
#define PYBIND11_MODULE(name, m) static void oracle_pybind_stub_##name(OracleFlowDummyModule& m)
