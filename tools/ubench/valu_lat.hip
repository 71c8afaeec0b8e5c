// micro-benchmark: issue cost (cycles) of the instruction patterns on the pivot chain of the front kernels, one wave
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double readlane_d(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
template <int MODE> __global__ void k(double *out, long long *cyc, double seed) {
    double a[16];
    for (int i = 0; i < 16; i++) a[i] = seed + threadIdx.x * 1e-3 + i;
    double x = seed * 0.5;
    long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < 64; it++) {
        if (MODE == 0) {           // 16 independent fmas
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = fma(a[i], x, 1e-9);
        } else if (MODE == 1) {    // 16 dependent fmas
#pragma unroll
            for (int i = 0; i < 16; i++) x = fma(x, 0.999, 1e-9);
        } else if (MODE == 2) {    // 16 x (readlane pair + fma with the scalar)
#pragma unroll
            for (int i = 0; i < 16; i++) { double s = readlane_d(a[(i + 1) & 15], i); a[i] = fma(a[i], s, 1e-9); }
        } else if (MODE == 3) {    // 16 readlane pairs first, then 16 fmas
            double s[16];
#pragma unroll
            for (int i = 0; i < 16; i++) s[i] = readlane_d(a[(i + 1) & 15], i);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = fma(a[i], s[i], 1e-9);
            __builtin_amdgcn_sched_barrier(0);
        } else if (MODE == 4) {    // dependent rsq + newton
            double y = __builtin_amdgcn_rsq(x + 2.0); double h = 0.5 * (x + 2.0);
            double e = fma(-h * y, y, 0.5); y = fma(y, e, y); e = fma(-h * y, y, 0.5); y = fma(y, e, y); x = y;
        } else if (MODE == 5) {    // 16 fmas with an SGPR operand that is constant
            double s = readlane_d(x, 3);
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = fma(a[i], s, 1e-9);
        } else if (MODE == 6) {    // LDS broadcast read + fma
            __shared__ double sh[64];
            sh[threadIdx.x & 63] = a[0];
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = fma(a[i], sh[i], 1e-9);
        }
    }
    long long t1 = clock64();
    double acc = x;
    for (int i = 0; i < 16; i++) acc += a[i];
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
    double *out; long long *cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8 * 8); hipMemset(cyc, 0, 64);
    for (int rep = 0; rep < 2; rep++) {
        k<0><<<1, 64>>>(out, cyc, 1.0); k<1><<<1, 64>>>(out, cyc, 1.0); k<2><<<1, 64>>>(out, cyc, 1.0); k<3><<<1, 64>>>(out, cyc, 1.0);
        k<4><<<1, 64>>>(out, cyc, 1.0); k<5><<<1, 64>>>(out, cyc, 1.0); k<6><<<1, 64>>>(out, cyc, 1.0);
    }
    hipDeviceSynchronize();
    long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const char *names[] = { "16 independent fma_f64", "16 dependent fma_f64", "16 x (2 readlane + fma)", "32 readlane then 16 fma", "rsq + 2 newton (dependent)", "readlane pair + 16 fma sgpr operand", "16 x (lds broadcast read + fma)" };
    for (int m = 0; m < 7; m++) printf("%-40s %8.1f cycles per iteration of 64\n", names[m], h[m] / 64.0);
    return 0;
}
