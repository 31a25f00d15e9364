// What the bf16 matrix pipe sustains under the socket power cap (gfx950): v_mfma_f32_32x32x16_bf16 on register operands only -- no LDS,
// no memory -- W wavefronts per SIMD on every CU, CHAINS independent accumulators per wavefront, for a few milliseconds, with random
// (non-zero) operands: executed PFLOP/s, matrix-pipe clocks per instruction and the shader clock the run settled at (s_memtime ticks
// against the 100 MHz wall clock).  A duty argument < 100 inserts idle s_sleep phases (what a kernel with exposed non-matrix phases
// looks like to the power controller).  Build + run: tools/microbench/run_mfma_power.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int CHAINS>
__global__ __launch_bounds__(256) void burn(const uint4 *seed, float *out, long long *ticks, long long *wall, int iters, int sleep_every, int sleep_len) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, seed[(tid * 8 + i) & 65535]);
        b[i] = __builtin_bit_cast(bf16x8, seed[(tid * 8 + 4 + i) & 65535]);
    }
    floatx16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = floatx16{0};
    const long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u + c) & 3], b[(u * 3 + c) & 3], acc[c], 0, 0, 0);
        if (sleep_every && (it % sleep_every) == sleep_every - 1)
            for (int s = 0; s < sleep_len; ++s) __builtin_amdgcn_s_sleep(16);
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][7];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0) {
        ticks[blockIdx.x] = t1 - t0;
        wall[blockIdx.x] = w1 - w0;
    }
}
int main(int argc, char **argv) {
    uint4 *seed; float *out; long long *ticks, *wall;
    std::vector<uint4> h(65536);
    srand(7);
    for (auto &v : h) {                       // random bf16 values in [0.5, 2): exponent 0x3F / 0x3F8.., random mantissas
        unsigned r[4];
        for (int i = 0; i < 4; ++i) r[i] = (0x3F00u | (rand() & 0xFF)) | ((0x3F00u | (rand() & 0xFF)) << 16);
        v = uint4{r[0], r[1], r[2], r[3]};
    }
    hipMalloc(&seed, h.size() * 16); hipMalloc(&out, 4); hipMalloc(&ticks, 4096 * 8); hipMalloc(&wall, 4096 * 8);
    hipMemcpy(seed, h.data(), h.size() * 16, hipMemcpyHostToDevice);
    const int cus = 256;
    struct Cfg { int wgs_per_cu, sleep_every, sleep_len; const char *name; };
    const Cfg cfgs[] = {{1, 0, 0, "1 wave/SIMD, 100 % duty"}, {2, 0, 0, "2 waves/SIMD, 100 % duty"}, {4, 0, 0, "4 waves/SIMD, 100 % duty"},
                        {4, 4, 24, "4 waves/SIMD, ~2/3 duty"}, {4, 4, 48, "4 waves/SIMD, ~1/2 duty"}};
    for (const Cfg &c : cfgs) {
        const int grid = cus * c.wgs_per_cu, iters = 6000 / c.wgs_per_cu;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {                   // the third run is reported: the power controller has settled
            hipEventRecord(e0);
            hipLaunchKernelGGL(burn<2>, dim3(grid), dim3(256), 0, 0, seed, out, ticks, wall, iters, c.sleep_every, c.sleep_len);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        std::vector<long long> t(grid), w(grid);
        hipMemcpy(t.data(), ticks, grid * 8, hipMemcpyDeviceToHost); hipMemcpy(w.data(), wall, grid * 8, hipMemcpyDeviceToHost);
        double tm = 0, wm = 0; for (int i = 0; i < grid; ++i) { tm += t[i]; wm += w[i]; } tm /= grid; wm /= grid;
        const double mfmas = (double)grid * 4 * iters * 8 * 2, flops = mfmas * 32768.0;
        printf("%-26s %7.2f ms  %6.3f PFLOP/s executed  shader clock %.0f MHz (s_memtime / wall)  %.1f clk per MFMA and SIMD\n", c.name, ms, flops / ms * 1e-12,
               tm / (wm / 100.0), tm / ((double)iters * 8 * 2 * c.wgs_per_cu));
    }
    return 0;
}
