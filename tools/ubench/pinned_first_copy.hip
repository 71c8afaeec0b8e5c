// What does the FIRST copy-engine access to a fresh pinned allocation cost?  hipcc --offload-arch=gfx950 -O2 tools/ubench/pinned_first_copy.hip -o tools/ubench/pinned_first_copy
// (round 6: the incremental demo's step 1 330 -- the step at which the append slack runs out and the pinned mirrors are reallocated -- spends 7.6 ms
// inside one hipMemcpyAsync whose copy itself takes 7 us)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k(double *p) { p[threadIdx.x] += 1.0; }
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double *d; hipMalloc((void **)&d, 8 << 20);
    for (int rep = 0; rep < 6; rep++) {
        const size_t bytes = (rep % 3 == 0) ? (64 << 10) : (rep % 3 == 1 ? (1 << 20) : (6 << 20));
        char *h; double t0 = now(); hipHostMalloc((void **)&h, bytes, hipHostMallocDefault); double t1 = now();
        memset(h, 0, bytes); double t2 = now();
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d);
        double a0 = now(); hipMemcpyAsync(h, d, 32 << 10, hipMemcpyDeviceToHost, s); double a1 = now();
        hipMemcpyAsync(h, d, 32 << 10, hipMemcpyDeviceToHost, s); double a2 = now();
        hipMemcpyAsync(d, h, 32 << 10, hipMemcpyHostToDevice, s); double a3 = now();
        hipStreamSynchronize(s); double a4 = now();
        printf("fresh pinned buffer of %zu KB: hipHostMalloc %.3f ms, memset %.3f | first D2H call %.3f ms, second %.3f, H2D %.3f, sync %.3f\n", bytes >> 10, t1 - t0, t2 - t1, a1 - a0, a2 - a1, a3 - a2, a4 - a3);
        if (rep >= 3) hipHostFree(h);
    }
    // ... and the first copies on a SECOND stream, and on the first stream with sizes not used before
    hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    char *h; hipHostMalloc((void **)&h, 8 << 20, hipHostMallocDefault); memset(h, 0, 8 << 20);
    for (int rep = 0; rep < 2; rep++) {
        double a0 = now(); hipMemcpyAsync(h, d, 32 << 10, hipMemcpyDeviceToHost, s2); double a1 = now();
        hipMemcpyAsync(d, h, 32 << 10, hipMemcpyHostToDevice, s2); double a2 = now();
        hipMemcpyAsync(d + (1 << 17), d, 32 << 10, hipMemcpyDeviceToDevice, s2); double a3 = now(); hipStreamSynchronize(s2);
        printf("second stream, pass %d: D2H call %.3f ms, H2D %.3f, D2D %.3f\n", rep, a1 - a0, a2 - a1, a3 - a2);
    }
    for (size_t nb : { (size_t)16, (size_t)4 << 10, (size_t)100 << 10, (size_t)3 << 20 }) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d);
        double a0 = now(); hipMemcpyAsync(h, d, nb, hipMemcpyDeviceToHost, s); double a1 = now();
        hipMemcpyAsync(d, h, nb, hipMemcpyHostToDevice, s); double a2 = now(); hipStreamSynchronize(s);
        printf("first stream, %zu bytes: D2H call %.3f ms, H2D %.3f\n", nb, a1 - a0, a2 - a1);
    }
    // ... and fresh DEVICE allocations: first copy out of / into one, first kernel touching one
    for (size_t mb : { (size_t)1, (size_t)8, (size_t)64, (size_t)1, (size_t)64 }) {
        double *dn; double t0 = now(); hipMalloc((void **)&dn, mb << 20); double t1 = now();
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, dn); double t2 = now(); hipStreamSynchronize(s); double t3 = now();
        double a0 = now(); hipMemcpyAsync(h, dn, 32 << 10, hipMemcpyDeviceToHost, s); double a1 = now();
        hipMemcpyAsync(dn, h, 32 << 10, hipMemcpyHostToDevice, s); double a2 = now(); hipStreamSynchronize(s); double a3 = now();
        double f0 = now(); hipFree(dn); double f1 = now();
        double b0 = now(); hipMemcpyAsync(h, d, 32 << 10, hipMemcpyDeviceToHost, s); double b1 = now(); hipStreamSynchronize(s);
        printf("fresh device buffer of %zu MB: hipMalloc %.3f ms, kernel launch %.3f + sync %.3f | D2H call %.3f, H2D %.3f, sync %.3f | hipFree %.3f | next D2H (old buffer) %.3f\n", mb, t1 - t0, t2 - t1, t3 - t2, a1 - a0, a2 - a1, a3 - a2, f1 - f0, b1 - b0);
    }
    return 0;
}
