// Board power of LDS operand reads alone (gfx950): every CU runs 8 wavefronts issuing ds_read_b128 back to back on random data for ~4 s while
// rocm-smi is sampled from the shell (tools/microbench/run_lds_power.sh); a second pass runs the same loop with the reads replaced by s_nop
// (clocked, idle wavefronts) for the baseline.  Prints the achieved LDS bytes per second.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const int tid = threadIdx.x;
    unsigned x = tid * 2654435761u + blockIdx.x;
    for (int i = tid; i < 16384; i += blockDim.x) { x = x * 1664525u + 1013904223u; lds[i] = x; }
    __syncthreads();
    float acc = 0.f;
    const int lane_off = (tid & 63) * 4 + (tid >> 6) * 256;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 1) {
                const int base = (lane_off + u * 1024 + it * 64) & 16383 & ~3;
                float4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(base * 4)); asm volatile("s_waitcnt lgkmcnt(8)"); acc += v.x;
            } else {
                asm volatile("s_nop 7");
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    if (acc == 12345.678f) out[0] = acc;
}
int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 1;
    const double seconds = argc > 2 ? atof(argv[2]) : 4.0;
    float *out; hipMalloc(&out, 4);
    const int iters = 20000;
    auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int r = 0; r < 4; ++r) {
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 65536, 0, out, iters);
            else hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 65536, 0, out, iters);
        }
        hipDeviceSynchronize(); launches += 4;
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double bytes = (double)launches * 256 * 512 * 16.0 * iters * 16;
    printf("mode %d: %.2f s, %.1f TB/s of LDS reads chip-wide (%.1f B/clk/CU at 2.4 GHz)\n", mode, dt, mode == 1 ? bytes / dt / 1e12 : 0.0, mode == 1 ? bytes / dt / 256 / 2.4e9 : 0.0);
    return 0;
}
