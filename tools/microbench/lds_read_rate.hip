// LDS read-rate microbenchmark (gfx950): bytes per clock per CU of ds_read_b32 / b64 / b128 with conflict-free, lane-linear
// addresses, W wavefronts per workgroup, one workgroup per CU.  Build + run: tools/microbench/run_lds_read_rate.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int BYTES>
__global__ __launch_bounds__(1024) void rd(float *out, long long *cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    float acc = 0.f;
    const int lane_off = (tid & 63) * (BYTES / 4) + (tid >> 6) * 64 * (BYTES / 4);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int base = (lane_off + u * 1024 + it * 64) & 16383 & ~(BYTES / 4 - 1);
            if (BYTES == 4) { float v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(base * 4)); asm volatile("s_waitcnt lgkmcnt(8)"); acc += v; }
            if (BYTES == 8) { float2 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(base * 4)); asm volatile("s_waitcnt lgkmcnt(8)"); acc += v.x; }
            if (BYTES == 16) { float4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(base * 4)); asm volatile("s_waitcnt lgkmcnt(8)"); acc += v.x; }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 12345.678f) out[0] = acc;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    for (int waves : {4, 8, 16}) {
        for (int bytes : {4, 8, 16}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (bytes == 4) hipLaunchKernelGGL(rd<4>, dim3(256), dim3(64 * waves), 65536, 0, out, cyc, iters);
                if (bytes == 8) hipLaunchKernelGGL(rd<8>, dim3(256), dim3(64 * waves), 65536, 0, out, cyc, iters);
                if (bytes == 16) hipLaunchKernelGGL(rd<16>, dim3(256), dim3(64 * waves), 65536, 0, out, cyc, iters);
                hipDeviceSynchronize();
            }
            std::vector<long long> h(256);
            hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
            double mean = 0; for (auto v : h) mean += v; mean /= 256;
            const double total = (double)iters * 16 * waves * 64 * bytes;
            printf("waves/WG %2d  ds_read_b%-3d  %.1f B/clk/CU  (%.2f clk per wave-instruction)\n", waves, bytes * 8, total / mean, mean / (iters * 16.0 * waves));
        }
    }
    return 0;
}
