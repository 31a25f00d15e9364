// Fused convolution epilogue: y = act(y + bias[c] (+ residual)), in place, NCHW, gfx950.
//
// The reference evaluates every backbone block as conv -> BatchNorm -> ReLU -> conv -> BatchNorm -> (+identity) -> ReLU
// (opencood/models/sub_modules/resblock.py:53-69), i.e. five element-wise passes over the activation per block on top
// of the two MIOpen convolutions.  In eval mode BatchNorm is an affine map per channel: its scale is folded into the
// convolution weights on the host (coalign_amd/backbone.py) and what remains -- per-channel shift, residual add, ReLU --
// is this single streaming pass (one read + one write, 16 B per lane), two per block instead of five.
#include "common.h"

namespace {

template <bool VEC4>
__global__ __launch_bounds__(256) void bias_act_kernel(float *__restrict__ y, const float *__restrict__ bias,
                                                       const float *__restrict__ res, int planes, int C, int HW, int relu) {
  for (int plane = blockIdx.y; plane < planes; plane += gridDim.y) {   // plane = n * C + c (grid.y is capped at 65535)
    const float b = bias ? bias[plane % C] : 0.f;
    const size_t base = (size_t)plane * HW;
    if constexpr (VEC4) {
        const int n4 = HW >> 2;
        float4 *yp = reinterpret_cast<float4 *>(y + base);
        const float4 *rp = reinterpret_cast<const float4 *>(res ? res + base : nullptr);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
            float4 v = yp[i];
            v.x += b; v.y += b; v.z += b; v.w += b;
            if (res) { const float4 r = rp[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            yp[i] = v;
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
            float v = y[base + i] + b;
            if (res) v += res[base + i];
            if (relu) v = fmaxf(v, 0.f);
            y[base + i] = v;
        }
    }
  }
}

}  // namespace

extern "C" int coalign_bias_act(float *y, const float *bias, const float *residual, int N, int C, int HW, int relu,
                                void *stream_) {
    using namespace coalign;
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || C <= 0 || HW <= 0) return COALIGN_ERR_BAD_SHAPE;
    if (N == 0) return COALIGN_OK;
    if (!y) return COALIGN_ERR_NULL_POINTER;
    if ((size_t)N * C > (size_t)INT32_MAX) return COALIGN_ERR_UNSUPPORTED;
    const bool vec = (HW % 4 == 0) && (((uintptr_t)y & 15) == 0) && (!residual || ((uintptr_t)residual & 15) == 0);
    const int per = vec ? HW / 4 : HW;
    int bx = (per + 255) / 256;
    if (bx > 64) bx = 64;
    const int planes = N * C;
    dim3 grid(bx, planes < 65535 ? planes : 65535);      // HIP's grid.y limit; the kernel strides over the planes beyond it
    if (vec) hipLaunchKernelGGL(bias_act_kernel<true>, grid, dim3(256), 0, stream, y, bias, residual, planes, C, HW, relu);
    else hipLaunchKernelGGL(bias_act_kernel<false>, grid, dim3(256), 0, stream, y, bias, residual, planes, C, HW, relu);
    return check_launch();
}

// Clear / fill a small device buffer with a KERNEL on the caller's stream (the per-frame counters of the post-processing buffers): inside a
// captured frame a hipMemsetAsync node is not reliably ordered against the kernel node behind it on ROCm 7.2 (common.h, fill_words), and a
// torch fill would be a library launch in the frame.
extern "C" int coalign_fill_words(void *p, size_t n_words, uint32_t value, void *stream) {
    if (!p && n_words) return COALIGN_ERR_NULL_POINTER;
    if (reinterpret_cast<uintptr_t>(p) & 3) return COALIGN_ERR_UNSUPPORTED;
    return coalign::fill_words(p, n_words, value, static_cast<hipStream_t>(stream));
}
