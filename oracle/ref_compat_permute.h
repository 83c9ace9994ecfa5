// ref_compat_permute.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// Force-included (-include) when compiling the reference's
// torchvision/csrc/ops/cpu/deform_conv2d_kernel.cpp against torch 2.10: that TU calls
// torch::stable::permute (cpu/deform_conv2d_kernel.cpp:699,828), which only exists in the
// stable ABI of newer torch (the reference targets 2.14, setup.py:154).  This supplies it
// through the dispatcher, the same way the reference's own StableABICompat.h wraps
// aten::sort / aten::index_select, without touching any reference source.
#pragma once
#include <torch/csrc/stable/c/shim.h>
#include <torch/csrc/stable/stableivalue_conversions.h>
#include <torch/csrc/stable/tensor.h>
#include <torch/headeronly/util/HeaderOnlyArrayRef.h>
#include <torch/headeronly/util/shim_utils.h>
#include <torch/headeronly/version.h>

#include <array>
#include <initializer_list>
#include <vector>

namespace torch {
namespace stable {
inline Tensor permute(const Tensor& self, std::initializer_list<int64_t> dims) {
  std::vector<int64_t> d(dims);
  std::array<StableIValue, 2> stack{
      torch::stable::detail::from(self),
      torch::stable::detail::from(torch::headeronly::IntHeaderOnlyArrayRef(d.data(), d.size()))};
  TORCH_ERROR_CODE_CHECK(torch_call_dispatcher("aten::permute", "", stack.data(), TORCH_ABI_VERSION));
  return torch::stable::detail::to<Tensor>(stack[0]);
}
}  // namespace stable
}  // namespace torch
