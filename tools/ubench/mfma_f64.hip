// micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950 (sets the roofline peak used in DESIGN.md)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f64.hip -o /tmp/mfma_f64 && /tmp/mfma_f64
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(double *out, int iters, double a0, double b0) {
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
    printf("%s: %d CUs, %d MHz\n", p.gcnArchName, cus, p.clockRate / 1000);
    for (int wpc = 1; wpc <= 2; wpc++) {                     // workgroups of 4 waves per CU
        const int grid = cus * wpc;
        float ms = timeit([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1.0); });
        double n = (double)grid * 4 * iters * 4;               // MFMA instructions
        printf("mfma_f64_16x16x4 x4 acc, %d WG/CU: %.3f ms  %.2f TFLOP/s  %.1f cycles/instr/SIMD @2.4GHz\n", wpc, ms, n * 2048 / ms / 1e9,
               ms * 1e-3 * 2.4e9 / ((double)iters * 4 * wpc));
        ms = timeit([&] { hipLaunchKernelGGL(k_mfma<16>, dim3(grid), dim3(256), 0, 0, out, iters / 4, 1.0, 1.0); });
        n = (double)grid * 4 * (iters / 4) * 16;
        printf("mfma_f64_16x16x4 x16 acc, %d WG/CU: %.3f ms  %.2f TFLOP/s\n", wpc, ms, n * 2048 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL(k_fma<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.000001, 1e-9); });
        n = (double)grid * 256 * iters * 8 * 2;
        printf("v_fma_f64 x8 acc, %d WG/CU: %.3f ms  %.2f TFLOP/s\n", wpc, ms, n / ms / 1e9);
    }
    return 0;
}
