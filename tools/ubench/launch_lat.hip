// Host <-> GPU round-trip latencies that bound a small incremental step (hipcc --offload-arch=gfx950 -O2 launch_lat.hip -o launch_lat):
//   a) one empty launch + hipStreamSynchronize            b) one empty launch, completion through a word in pinned host memory (host spins)
//   c) three empty launches + hipStreamSynchronize        d) a RESIDENT kernel polling a word in pinned host memory: host rings, kernel answers
//   e) as b) with the kernel first reading 1 KiB of pinned host memory (the patch payload of a step)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_empty() {}
__global__ void k_flag(volatile int *done, int seq) { if (threadIdx.x == 0) { __hip_atomic_store((int *)done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); } }
__global__ void k_read_flag(const int *src, int *dst, volatile int *done, int seq) {
    dst[threadIdx.x] = src[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence_system(); __hip_atomic_store((int *)done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
// resident worker: answers every ring until told to stop or idle for too long (never spins for ever)
__global__ void k_resident(volatile int *ring, volatile int *done, const int *src, int *dst, long long max_idle_cycles) {
    int last = 0; long long t0 = wall_clock64();
    for (;;) {
        int r = __hip_atomic_load((int *)ring, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (r < 0) return;
        if (r != last) {
            last = r;
            if (src) dst[threadIdx.x] = src[threadIdx.x];
            __syncthreads();
            if (threadIdx.x == 0) { __threadfence_system(); __hip_atomic_store((int *)done, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
            t0 = wall_clock64();
        } else if (wall_clock64() - t0 > max_idle_cycles) return;
        __builtin_amdgcn_s_sleep(1);
    }
}
static void report(const char *name, std::vector<double> &v) {
    std::sort(v.begin(), v.end());
    printf("%-64s median %7.2f us   p10 %7.2f   p90 %7.2f\n", name, v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10]);
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int *h; CK(hipHostMalloc(&h, 4096 + 1024, hipHostMallocDefault));
    volatile int *ring = h, *done = h + 16; int *src = h + 1024;
    int *d; CK(hipMalloc(&d, 4096));
    const int n = 2000;
    std::vector<double> v(n);
    for (int i = 0; i < 200; i++) { hipLaunchKernelGGL(k_empty, 1, 64, 0, s); CK(hipStreamSynchronize(s)); }
    for (int i = 0; i < n; i++) { double t = now_us(); hipLaunchKernelGGL(k_empty, 1, 64, 0, s); CK(hipStreamSynchronize(s)); v[i] = now_us() - t; }
    report("a) 1 launch + hipStreamSynchronize", v);
    for (int i = 0; i < n; i++) { double t = now_us(); hipLaunchKernelGGL(k_empty, 1, 64, 0, s); v[i] = now_us() - t; CK(hipStreamSynchronize(s)); }
    report("   (host time of the launch call alone)", v);
    *done = 0;
    for (int i = 0; i < n; i++) { double t = now_us(); hipLaunchKernelGGL(k_flag, 1, 64, 0, s, done, i + 1); while (*done != i + 1) {} v[i] = now_us() - t; }
    CK(hipStreamSynchronize(s));
    report("b) 1 launch, completion word in pinned memory (host spins)", v);
    for (int i = 0; i < n; i++) { double t = now_us(); for (int k = 0; k < 3; k++) hipLaunchKernelGGL(k_empty, 1, 64, 0, s); CK(hipStreamSynchronize(s)); v[i] = now_us() - t; }
    report("c) 3 launches + hipStreamSynchronize", v);
    *done = 0;
    for (int i = 0; i < n; i++) { double t = now_us(); for (int k = 0; k < 2; k++) hipLaunchKernelGGL(k_empty, 1, 64, 0, s); hipLaunchKernelGGL(k_flag, 1, 64, 0, s, done, i + 1); while (*done != i + 1) {} v[i] = now_us() - t; }
    CK(hipStreamSynchronize(s));
    report("   3 launches, completion word in pinned memory", v);
    *done = 0;
    for (int i = 0; i < n; i++) { double t = now_us(); src[0] = i; hipLaunchKernelGGL(k_read_flag, 1, 256, 0, s, src, d, done, i + 1); while (*done != i + 1) {} v[i] = now_us() - t; }
    CK(hipStreamSynchronize(s));
    report("e) 1 launch reading 1 KiB of pinned memory, completion word", v);
    // resident worker: exits by itself after 0.2 s without a ring (100 MHz wall clock)
    *ring = 0; *done = 0;
    hipLaunchKernelGGL(k_resident, 1, 256, 0, s, ring, done, (const int *)nullptr, d, 20000000ll);
    for (int i = 0; i < n; i++) { double t = now_us(); *ring = i + 1; while (*done != i + 1) {} v[i] = now_us() - t; }
    *ring = -1; CK(hipStreamSynchronize(s));
    report("d) resident kernel: ring -> answer through pinned memory", v);
    *ring = 0; *done = 0;
    hipLaunchKernelGGL(k_resident, 1, 256, 0, s, ring, done, (const int *)src, d, 20000000ll);
    for (int i = 0; i < n; i++) { double t = now_us(); src[0] = i; *ring = i + 1; while (*done != i + 1) {} v[i] = now_us() - t; }
    *ring = -1; CK(hipStreamSynchronize(s));
    report("   ... reading 1 KiB of pinned memory per ring", v);
    return 0;
}
