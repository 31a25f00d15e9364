// Shared host-side helpers for the C-ABI entry points (gfx950 only, no portability layers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "coalign_amd.h"

namespace coalign {

void set_hip_error(hipError_t e);  // records the error string for coalign_last_hip_error() (thread local)

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_hip_error(e);
        return COALIGN_ERR_HIP;
    }
    return COALIGN_OK;
}

inline int hip_call(hipError_t e) {
    if (e != hipSuccess) {
        set_hip_error(e);
        return COALIGN_ERR_HIP;
    }
    return COALIGN_OK;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Laboratory switches (ablations, alternative tile geometries, measurement aids).  The PRODUCT library reads none of them: lab_env() is the compile-time
// default there.  Only the laboratory build (`python -m coalign_amd.build --lab`, -DCOALIGN_LAB, loaded with COALIGN_LAB=1 by tools/ and by the tests that
// compare variants) looks at the environment.  tests/test_host_cpu.py checks the product library's strings.
#ifdef COALIGN_LAB
inline int lab_env(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
#else
constexpr int lab_env(const char *, int dflt) { return dflt; }
#endif

// XCD-aware, bijective block remap (guide T1): blocks are dispatched round-robin over the 8 XCDs, so give
// each XCD a contiguous chunk of the logical index space -> neighbouring tiles share one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    constexpr int NX = 8;
    const int xcd = bid % NX, slot = bid / NX;
    const int q = nblocks / NX, r = nblocks % NX;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// Streaming ("non-temporal") stores for outputs the NEXT launch reads (from any XCD, i.e. through memory anyway): the line leaves the XCD's write-back L2 as it is
// produced instead of waiting dirty for the end-of-kernel release, which then has megabytes to write back before the launch can complete.  Round 6, measured
// (profiles/round6/experiments/streaming_stores.txt): pillar rows 13.4-13.9 -> 12.9 us, fused maps 20.4-21.0 -> 18.7-19.4 us (adopted); the convolutions' SplitMap
// output -4 % alone on the GPU but -1.7 % frames/s inside the two-stream pipeline, their channels-last output +16 % (16-byte pieces of lines): not adopted.
__device__ __forceinline__ void store_stream(float *p, float v) { __builtin_nontemporal_store(v, p); }
typedef float coalign_f4 __attribute__((ext_vector_type(4)));
typedef unsigned coalign_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_stream(float4 *p, float4 v) { __builtin_nontemporal_store(coalign_f4{v.x, v.y, v.z, v.w}, reinterpret_cast<coalign_f4 *>(p)); }
__device__ __forceinline__ void store_stream(uint4 *p, uint4 v) { __builtin_nontemporal_store(coalign_u4{v.x, v.y, v.z, v.w}, reinterpret_cast<coalign_u4 *>(p)); }

// Order a wavefront's own LDS traffic: write by some lanes, read by others of the SAME wave.  The LDS unit executes one
// wave's DS instructions in issue order, so no hardware wait is needed -- only the compiler must not reorder across this
// point.  (A __builtin_amdgcn_fence here can also drain vmcnt, i.e. wait for every prefetched global load.)
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// ---- sp16, the "split pair": an fp32 value on the fp16 matrix cores (round 5; csrc/conv3x3_emu.hip, csrc/conv3x3_sp.hip) ------------------------------
// x~ = x rounded to 22 significant bits (ties away from zero) = h + 2^-10 l, where h = the leading 11 bits of x~ as an fp16 number (an exact truncation) and
// l = (x~ - h) * 2^10 = the following 11 bits, scaled by 2^10 so that they are a NORMAL fp16 number whenever |x| >= 2^-14 (unscaled, x - h is subnormal below
// |x| = 2^-3 and what it keeps shrinks to an absolute 2^-25: round 4's precision hole).  Exact for 2^-14 <= |x| <= 65504; smaller values keep an absolute
// 2^-34; larger ones clamp (h and l saturate at 65504: x~ <= 65504 + 63.97) -- finite, and reported by the producing kernels' range word where they have one.
// A pair is canonical: splitting x~ again returns the same (h, l), so a kernel that splits fp32 inputs itself and a kernel that reads stored pairs see the
// same operands bit for bit.  In a product sum_k w_k x_k the terms w_h x_h go to one fp32 accumulator, w_h x_l + w_l x_h (both carrying 2^10) to a second one
// that enters with 2^-10 at the end; w_l x_l (< 2^-20 |w x|, zero mean: the weights' h is rounded to nearest on the host) is dropped.
constexpr float kSp16LowScale = 1024.f, kSp16LowInv = 1.f / 1024.f;

__device__ __forceinline__ float sp16_round(float x) { return __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, x) + 2u) & ~3u); }

// two values -> (h0 | h1 << 16, l0 | l1 << 16), the operand order of the 16-bit matrix instructions
__device__ __forceinline__ void sp16_split2(float a, float b, unsigned &hi, unsigned &lo) {
    const float ar = sp16_round(a), br = sp16_round(b);
    const auto h = __builtin_amdgcn_cvt_pkrtz(ar, br);
    // (x~ - h as fmaf(h, -1, x~): the same correctly rounded difference, but ONE v_fma_mix_f32 reading h's half of the packed register instead of a conversion + a subtraction)
    const auto l = __builtin_amdgcn_cvt_pkrtz(fmaf((float)h[0], -1.0f, ar) * kSp16LowScale, fmaf((float)h[1], -1.0f, br) * kSp16LowScale);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

__device__ __forceinline__ float sp16_join(_Float16 h, _Float16 l) { return fmaf((float)l, kSp16LowInv, (float)h); }      // exact

// Workspace / flag clearing as a KERNEL instead of hipMemsetAsync.  Under HIP-graph replay (ROCm 7.2) a memset node followed by a kernel
// that polls or accumulates into the cleared words was observed to race from an idle GPU (wrong stream-K hand-overs in the first replays
// after a device synchronise, tools/diag_graph.py); a kernel node is ordered like every other kernel of the capture.
static __global__ void fill_words_kernel(unsigned *__restrict__ p, size_t n_words, unsigned value) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) p[i] = value;
}

static __global__ void fill_vec4_kernel(uint4 *__restrict__ p, size_t n_vec, unsigned value) {      // 16 B per lane, coalesced
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_vec) p[i] = make_uint4(value, value, value, value);
}

inline int fill_words(void *p, size_t n_words, unsigned value, hipStream_t stream) {
    if (n_words == 0) return COALIGN_OK;
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0 && n_words >= 1024) {
        const size_t n_vec = n_words / 4;
        hipLaunchKernelGGL(fill_vec4_kernel, dim3((unsigned)((n_vec + 255) / 256)), dim3(256), 0, stream, static_cast<uint4 *>(p), n_vec, value);
        int rc = check_launch();
        if (rc) return rc;
        p = static_cast<unsigned *>(p) + n_vec * 4;
        n_words -= n_vec * 4;
        if (n_words == 0) return COALIGN_OK;
    }
    hipLaunchKernelGGL(fill_words_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, stream, static_cast<unsigned *>(p), n_words, value);
    return check_launch();
}

}  // namespace coalign
