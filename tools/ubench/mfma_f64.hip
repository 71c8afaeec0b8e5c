// micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950 (sets the roofline peak used in DESIGN.md)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f64.hip -o /tmp/mfma_f64 && /tmp/mfma_f64
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
// clk[0..1]: shader-clock ticks (s_memtime) and constant 100 MHz ticks (s_memrealtime) spent inside the kernel by workgroup 0:
// their ratio is the clock the compute units actually ran at UNDER THIS LOAD -- the "cycles per instruction" below use it
template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(double *out, int iters, double a0, double b0, long long *clk = nullptr) {
    const long long c0 = clock64(), w0 = wall_clock64();
    d4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (d4){ 0, 0, 0, 0 };
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}
template <int NACC>
__global__ void __launch_bounds__(256) k_fma(double *out, int iters, double a0, double b0) {
    double acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = fma(acc[i], a0, b0);
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i];
    if (s == 12345.678) out[0] = s;
}
template <class F> static float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    double *out; hipMalloc(&out, 8);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 20000;
    printf("%s: %d CUs, %d MHz (device property)\n", p.gcnArchName, cus, p.clockRate / 1000);
    long long *clk; hipMalloc(&clk, 16); long long hclk[2];
    for (int wpc = 1; wpc <= 2; wpc++) {                     // workgroups of 4 waves per CU
        const int grid = cus * wpc;
        float ms = timeit([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1.0, clk); });
        hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
        const double mhz = 100.0 * (double)hclk[0] / (double)hclk[1];      // measured shader clock under FP64 MFMA load
        double n = (double)grid * 4 * iters * 4;               // MFMA instructions
        printf("mfma_f64_16x16x4 x4 acc, %d WG/CU: %.3f ms  %.2f TFLOP/s  measured clock %.0f MHz (s_memtime / s_memrealtime)  %.1f cycles/instr/SIMD at that clock  "
               "-> data-sheet rate (64 cycles/instr) at that clock would be %.1f TFLOP/s\n", wpc, ms, n * 2048 / ms / 1e9, mhz,
               ms * 1e-3 * mhz * 1e6 / ((double)iters * 4 * wpc), 2048.0 * 4 * cus * mhz * 1e6 / 64 / 1e12);
        ms = timeit([&] { hipLaunchKernelGGL(k_mfma<16>, dim3(grid), dim3(256), 0, 0, out, iters / 4, 1.0, 1.0); });
        n = (double)grid * 4 * (iters / 4) * 16;
        printf("mfma_f64_16x16x4 x16 acc, %d WG/CU: %.3f ms  %.2f TFLOP/s\n", wpc, ms, n * 2048 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL(k_fma<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.000001, 1e-9); });
        n = (double)grid * 256 * iters * 8 * 2;
        printf("v_fma_f64 x8 acc, %d WG/CU: %.3f ms  %.2f TFLOP/s\n", wpc, ms, n / ms / 1e9);
    }
    // occupancy x accumulator sweep (round 4): what the wide-update kernel can hope for at 3-5 waves per SIMD
    for (int wpc = 1; wpc <= 6; wpc++) {
        const int grid = cus * wpc;
        float m1 = timeit([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1.0, (long long *)nullptr); });
        float m2 = timeit([&] { hipLaunchKernelGGL(k_mfma<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1.0, (long long *)nullptr); });
        float m4 = timeit([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1.0, (long long *)nullptr); });
        float m8 = timeit([&] { hipLaunchKernelGGL(k_mfma<8>, dim3(grid), dim3(256), 0, 0, out, iters / 2, 1.0, 1.0, (long long *)nullptr); });
        const double n = (double)grid * 4 * iters * 2048 / 1e9;
        printf("sweep %d waves/SIMD: acc x1 %.1f  x2 %.1f  x4 %.1f  x8 %.1f TFLOP/s\n", wpc, n / m1, n * 2 / m2, n * 4 / m4, n * 4 / m8);
    }
    return 0;
}
