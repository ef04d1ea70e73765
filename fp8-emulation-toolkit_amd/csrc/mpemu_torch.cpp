// mpemu_torch.cpp -- the thin PyTorch-ROCm extension over the C ABI (include/fpemu_hip.h): the operator boundary's
// two hot calls, FPEmuOp.apply (reference mpemu/pytquant/hip/fpemu.py:43-71 -> hip/fpemu_impl.cpp:27-37) and
// quantize_tensor's per-tensor "max" branch (reference qutils.py:282-287), with the argument checks, the output
// allocation, the stream / workspace lookup and the straight-through autograd node done in C++ instead of Python +
// ctypes.  Per-call host time is what the hooked models of BASELINE configs 3 / 4 are bound by (73 casts per
// BERT-base forward, ~640 per ResNet-50 step).  No arithmetic lives here: every number comes out of libfpemu_hip.so.
//
// torch supplies device memory (empty_like through the caching allocator), the current HIP stream and autograd;
// nothing else.  There is no CPU path: a CPU tensor raises.
#include <torch/extension.h>

// torch-ROCm presents HIP devices as device type "cuda": the masquerading guard / stream types are the ones to use
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <mutex>
#include <unordered_map>

#include "../../include/fpemu_hip.h"

namespace {

int dtype_id(const at::Tensor &t) {
  switch (t.scalar_type()) {
    case at::kFloat: return FPEMU_F32;
    case at::kHalf: return FPEMU_F16;
    case at::kBFloat16: return FPEMU_BF16;
    case at::kDouble: return FPEMU_F64;   // plain casts only (fpemu_cast); the library refuses it elsewhere
    default:
      TORCH_CHECK_TYPE(false, "fpemu: unsupported tensor dtype ", t.scalar_type(), " (float32, float64, float16, bfloat16 only)");
  }
}

void require_device(const at::Tensor &t, const char *what) {
  TORCH_CHECK(t.is_cuda(), "fpemu: ", what, " must live on a ROCm device (got ", t.device(), "); this build has no CPU path");
  TORCH_CHECK(t.is_contiguous(), "fpemu: ", what, " must be contiguous");  // hip/fpemu_impl.cpp:23-25
}

void check(int rc) {
  if (rc == 0) return;
  const char *msg = fpemu_last_error();
  if (rc < 0 && rc != FPEMU_EUNSUPPORTED) TORCH_CHECK_VALUE(false, msg);
  TORCH_CHECK(false, rc > 0 ? "HIP error: " : "", msg);
}

// Scratch of the per-tensor path: one 257-word buffer per (device, stream), as in _binding._stream_workspace.
// Never cached while the stream is capturing (the buffer would belong to that graph's pool).
std::mutex g_ws_mutex;
std::unordered_map<uint64_t, at::Tensor> g_ws;
constexpr int64_t kWsWords = 257;

at::Tensor stream_workspace(const at::Tensor &like, const c10::hip::HIPStreamMasqueradingAsCUDA &stream) {
  const uint64_t key = ((uint64_t)(uint8_t)like.get_device() << 56) ^ (uint64_t)(uintptr_t)stream.stream();
  {
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    auto it = g_ws.find(key);
    if (it != g_ws.end()) return it->second;
  }
  at::Tensor ws = at::empty({kWsWords}, like.options().dtype(at::kInt));
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream.stream(), &st) == hipSuccess && st == hipStreamCaptureStatusNone) {
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    g_ws.emplace(key, ws);
  }
  return ws;
}

at::Tensor cast_impl(const at::Tensor &input, bool inplace, int64_t fmt, int64_t rmode, double scale, uint64_t seed,
                     uint64_t offset) {
  TORCH_CHECK(input.is_cuda(), "fpemu: input must live on a ROCm device (got ", input.device(), "); this build has no CPU path");
  at::Tensor src = input.contiguous();
  at::Tensor dst = inplace ? src : at::empty_like(src);
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(src.device());
  const auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(src.get_device());
  check(fpemu_cast(src.data_ptr(), dst.data_ptr(), (uint64_t)src.numel(), dtype_id(src), (int)fmt, (int)rmode, (float)scale,
                   seed, offset, 0, (void *)stream.stream()));
  if (inplace && !src.is_same(input)) input.copy_(src);  // a non-contiguous tensor cast "in place"
  return inplace ? input : dst;
}

at::Tensor quantize_per_tensor_impl(const at::Tensor &input, bool inplace, int64_t fmt, int64_t rmode, double flt_max,
                                    bool clamp, uint64_t seed, uint64_t offset, const c10::optional<at::Tensor> &ws_opt) {
  TORCH_CHECK(input.is_cuda(), "fpemu: input must live on a ROCm device (got ", input.device(), "); this build has no CPU path");
  at::Tensor src = input.contiguous();
  at::Tensor dst = inplace ? src : at::empty_like(src);
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(src.device());
  const auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(src.get_device());
  // an override workspace that lives on another device is ignored, as _binding._stream_workspace does (a graph
  // captured for one GPU must not stop tensors on a second one from being quantised)
  const bool use_override = ws_opt.has_value() && ws_opt->defined() && ws_opt->is_cuda() &&
                            ws_opt->get_device() == src.get_device();
  at::Tensor ws = use_override ? *ws_opt : stream_workspace(src, stream);
  TORCH_CHECK(ws.is_cuda() && ws.scalar_type() == at::kInt && ws.numel() >= 1 && ws.is_contiguous(),
              "fpemu: workspace must be a contiguous int32 tensor");
  check(fpemu_quantize_per_tensor_ws(src.data_ptr(), dst.data_ptr(), (uint64_t)src.numel(), dtype_id(src), (int)fmt,
                                     (int)rmode, (float)flt_max, clamp ? 1 : 0, (uint32_t *)ws.data_ptr(),
                                     (uint64_t)ws.numel(), seed, offset, (void *)stream.stream()));
  if (inplace && !src.is_same(input)) input.copy_(src);
  return inplace ? input : dst;
}

// quantize_tensor's per-channel / fine-grained "max" branches (reference qutils.py:289-337: a Python loop of per-row
// casts that never goes through autograd): one launch of the rows kernel over the [rows, numel / rows] view.
// offset: first of the `rows` consecutive call offsets a loop of per-row casts would draw (stochastic modes).
at::Tensor quantize_rows(const at::Tensor &input, bool inplace, int64_t fmt, int64_t rmode, double flt_max, int64_t rows,
                         bool clamp, uint64_t seed, uint64_t offset) {
  TORCH_CHECK(input.is_cuda(), "fpemu: input must live on a ROCm device (got ", input.device(), "); this build has no CPU path");
  TORCH_CHECK(rows > 0 && input.numel() % rows == 0, "fpemu: ", input.numel(), " elements do not split into ", rows, " rows");
  at::Tensor src = input.contiguous();
  at::Tensor dst = inplace ? src : at::empty_like(src);
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(src.device());
  const auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(src.get_device());
  check(fpemu_quantize_rows(src.data_ptr(), dst.data_ptr(), (uint64_t)rows, (uint64_t)(src.numel() / rows), dtype_id(src),
                            (int)fmt, (int)rmode, (float)flt_max, clamp ? 1 : 0, seed, offset, (void *)stream.stream()));
  if (inplace && !src.is_same(input)) input.copy_(src);
  return inplace ? input : dst;
}

// Straight-through estimator (reference pytquant/cpp/fpemu.py:68-71): the gradient passes unchanged.
struct CastSTE : public torch::autograd::Function<CastSTE> {
  static at::Tensor forward(torch::autograd::AutogradContext *ctx, const at::Tensor &input, bool inplace, int64_t fmt,
                            int64_t rmode, double scale, int64_t seed, int64_t offset) {
    at::Tensor out = cast_impl(input, inplace, fmt, rmode, scale, (uint64_t)seed, (uint64_t)offset);
    if (inplace) ctx->mark_dirty({input});
    return out;
  }
  static torch::autograd::variable_list backward(torch::autograd::AutogradContext *, torch::autograd::variable_list grads) {
    return {grads[0], at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
  }
};

struct QuantizeSTE : public torch::autograd::Function<QuantizeSTE> {
  static at::Tensor forward(torch::autograd::AutogradContext *ctx, const at::Tensor &input, bool inplace, int64_t fmt,
                            int64_t rmode, double flt_max, bool clamp, int64_t seed, int64_t offset,
                            const c10::optional<at::Tensor> &ws) {
    at::Tensor out = quantize_per_tensor_impl(input, inplace, fmt, rmode, flt_max, clamp, (uint64_t)seed, (uint64_t)offset, ws);
    if (inplace) ctx->mark_dirty({input});
    return out;
  }
  static torch::autograd::variable_list backward(torch::autograd::AutogradContext *, torch::autograd::variable_list grads) {
    return {grads[0], at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(),
            at::Tensor()};
  }
};

bool needs_graph(const at::Tensor &t) { return at::GradMode::is_enabled() && t.requires_grad(); }

// FPEmuOp.apply(tensor, mode, inplace, scale) for a dense tensor and a plain cast (no block norm)
at::Tensor fpemu_op(const at::Tensor &input, bool inplace, int64_t fmt, int64_t rmode, double scale, uint64_t seed,
                    uint64_t offset) {
  if (needs_graph(input)) return CastSTE::apply(input, inplace, fmt, rmode, scale, (int64_t)seed, (int64_t)offset);
  return cast_impl(input, inplace, fmt, rmode, scale, seed, offset);
}

// quantize_tensor's per-tensor "max" branch: amax -> device-side scale -> cast, no host sync
at::Tensor quantize_per_tensor(const at::Tensor &input, bool inplace, int64_t fmt, int64_t rmode, double flt_max, bool clamp,
                               uint64_t seed, uint64_t offset, const c10::optional<at::Tensor> &ws) {
  if (needs_graph(input))
    return QuantizeSTE::apply(input, inplace, fmt, rmode, flt_max, clamp, (int64_t)seed, (int64_t)offset, ws);
  return quantize_per_tensor_impl(input, inplace, fmt, rmode, flt_max, clamp, seed, offset, ws);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "thin PyTorch-ROCm front of libfpemu_hip.so (no arithmetic here)";
  m.def("abi_version", []() { return fpemu_abi_version(); });
  m.def("fpemu_op", &fpemu_op, py::arg("input"), py::arg("inplace"), py::arg("fmt"), py::arg("rmode"), py::arg("scale"),
        py::arg("seed"), py::arg("offset"));
  m.def("quantize_per_tensor", &quantize_per_tensor, py::arg("input"), py::arg("inplace"), py::arg("fmt"), py::arg("rmode"),
        py::arg("flt_max"), py::arg("clamp"), py::arg("seed"), py::arg("offset"), py::arg("workspace") = py::none());
  m.def("quantize_rows", &quantize_rows, py::arg("input"), py::arg("inplace"), py::arg("fmt"), py::arg("rmode"),
        py::arg("flt_max"), py::arg("rows"), py::arg("clamp"), py::arg("seed"), py::arg("offset"));
}
