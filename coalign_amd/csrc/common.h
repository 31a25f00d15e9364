// Shared host-side helpers for the C-ABI entry points (gfx950 only, no portability layers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

}  // namespace coalign
