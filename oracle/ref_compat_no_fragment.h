// ref_compat_no_fragment.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// Force-included (-include) when compiling the reference's torchvision/csrc/ops/quantized/cpu/{qnms,qroi_align}_kernel.cpp:
// besides their CPU-key implementation those two TUs DEFINE their schema in a STABLE_TORCH_LIBRARY_FRAGMENT(torchvision, m)
// block (qnms_kernel.cpp:148-150, qroi_align_kernel.cpp:234-237).  vision_amd's dispatcher glue owns the schema definitions
// (one definition per process), so here the block becomes an ordinary unused static function: the m.def(...) inside still
// compiles, it is just never run.  No reference source is modified.
#pragma once
#include <torch/csrc/stable/library.h>

#undef STABLE_TORCH_LIBRARY_FRAGMENT
#define STABLE_TORCH_LIBRARY_FRAGMENT(ns, m) \
  [[maybe_unused]] static void tvmi_ref_unused_fragment_##ns(torch::stable::detail::StableLibrary& m)
