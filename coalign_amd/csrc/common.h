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

// Order a wavefront's own LDS traffic: write by some lanes, read by others of the SAME wave.  The LDS unit executes one
// wave's DS instructions in issue order, so no hardware wait is needed -- only the compiler must not reorder across this
// point.  (A __builtin_amdgcn_fence here can also drain vmcnt, i.e. wait for every prefetched global load.)
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

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
