// micro-benchmark: do FP64 MFMAs and FP64 VALU fmas of DIFFERENT waves on the same SIMD overlap on this part (MI355X: data-sheet FP64 vector = FP64
// matrix = 78.6 TFLOP/s)?  One workgroup of 8 waves = 2 per SIMD: waves 0-3 issue back-to-back v_mfma_f64_16x16x4_f64 (4 accumulators), waves 4-7
// independent v_fma_f64 (16 accumulators); each group alone, then both together.  Cycles of the slower group per round.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
template <int MODE> __global__ void __launch_bounds__(512) k(double *out, long long *cyc, double seed) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = wave < 4 && (MODE & 1), do_fma = wave >= 4 && (MODE & 2);
    d4_t acc[4]; double a[16];
    for (int i = 0; i < 4; i++) acc[i] = (d4_t){ seed, seed, seed, seed };
    for (int i = 0; i < 16; i++) a[i] = seed + i + threadIdx.x * 1e-3;
    const double x = seed * 0.5, y = seed * 0.25;
    __syncthreads();
    const long long t0 = clock64();
    if (do_mfma) {
#pragma unroll 1
        for (int it = 0; it < 256; it++) {
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
        }
    }
    if (do_fma) {
#pragma unroll 1
        for (int it = 0; it < 256; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = fma(a[i], x, 1e-9);
        }
    }
    const long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; i++) s += a[i];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[MODE * 8 + wave] = t1 - t0;
}
int main() {
    double *out; long long *cyc;
    hipMalloc(&out, 512 * 8); hipMalloc(&cyc, 4 * 8 * 8); hipMemset(cyc, 0, 256);
    for (int rep = 0; rep < 2; rep++) { k<1><<<1, 512>>>(out, cyc, 1.0); k<2><<<1, 512>>>(out, cyc, 1.0); k<3><<<1, 512>>>(out, cyc, 1.0); }
    hipDeviceSynchronize();
    long long h[32]; hipMemcpy(h, cyc, 256, hipMemcpyDeviceToHost);
    printf("MFMA waves alone: %lld cycles for 1024 MFMAs per wave (%.1f per MFMA) = %.1f flop/cycle/SIMD\n", h[8 + 0], h[8 + 0] / 1024.0, 2048.0 * 1024 / h[8 + 0]);
    printf("fma  waves alone: %lld cycles for 4096 v_fma_f64 per wave (%.2f per fma) = %.1f flop/cycle/SIMD\n", h[16 + 4], h[16 + 4] / 4096.0, 128.0 * 4096 / h[16 + 4]);
    printf("both together:    MFMA waves %lld cycles (%.1f per MFMA), fma waves %lld cycles (%.2f per fma) -> %.1f flop/cycle/SIMD combined over the longer of the two\n",
           h[24 + 0], h[24 + 0] / 1024.0, h[24 + 4], h[24 + 4] / 4096.0, (2048.0 * 1024 + 128.0 * 4096) / (double)(h[24 + 0] > h[24 + 4] ? h[24 + 0] : h[24 + 4]));
    return 0;
}
