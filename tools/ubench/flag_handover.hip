// The flag hand-over of the multi-level launches, alone (hipcc --offload-arch=gfx950 -O2 flag_handover.hip -o flag_handover):
// a binary tree of 255 workgroups (128 leaves, 8 levels -- the shape of the chain-like graphs whose solves differed once in 10^4,
// profiles/r05_flag_soak.txt), dependencies with lower workgroup ids, two kernels per round exactly as in the solver:
//   A  ("k_linearize"): prepares the flags for the round, plus some unrelated stores from other workgroups
//   B  ("k_front_small", multi-level): every workgroup waits for its two children's flags, reads their 256 data words with device-scope
//      loads and checks them against the round number, then writes its own 256 words (plain stores, four waves), publishes its flag and
//      leaves a marker behind.
// Protocols:
//   reset = 0  flags are 0 / 1, reset by PLAIN stores in A, a wait passes on any non-zero value        (the solver before round 5's change)
//   reset = 1  the same, reset by device-scope stores
//   reset = 2  flags carry the round number (a counter in device memory that A advances), never reset  (the solver now)
//   wait  = 1  every wave waits for its own stores (s_waitcnt vmcnt(0)) before the publishing barrier
//   release = 0  the flag's release as the compiler emitted it at 9 of the solver's 24 sites before the fix: buffer_wbl2 and the store with NO
//                s_waitcnt between them;  1 = write-back, wait, store
// A wrong data word is classified by the child's marker: marker != round -> the wait passed before the child published in this round
// ("early"); marker == round -> the flag was right and the data word was not ("stale data").  A counter that reads back old is "stale epoch".
//   usage: flag_handover [seconds per configuration] [configurations: the first n of {before the fix, after, ...}]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
constexpr int NWG = 255, NLEAF = 128, W = 256;
constexpr int EPOCH0 = 1 << 20;
struct Counts { unsigned long long early, stale_data, stale_epoch, timeout, handovers; };

__device__ __forceinline__ int ld_dev(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void __launch_bounds__(256) k_prepare(int *flags, int *epoch, int *scratch, int reset, int round) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (reset == 2) { if (t == 0) atomicAdd(epoch, 1); }
    else if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < NWG; i += blockDim.x) {
            if (reset == 0) flags[i] = 0;
            else __hip_atomic_store(flags + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    scratch[t] = round;                 // (the other stores of the kernel before)
}

__global__ void __launch_bounds__(W) k_tree(int *flags, const int *epoch, int *data, int *marker, int reset, int wait_stores, int release, int round, Counts *cnt) {
    const int t = blockIdx.x, tid = threadIdx.x;
    // children of node t in a bottom-up numbering: leaves 0..127, then 64 nodes 128..191 with children (2 k, 2 k + 1), ...
    int c0 = -1, c1 = -1;
    if (t >= NLEAF) {
        int base = 0, n = NLEAF, first = NLEAF;           // level l: n nodes starting at base; their parents: n / 2 nodes starting at first
        while (t >= first + n / 2) { base = first; first += n / 2; n /= 2; }
        const int k = t - first; c0 = base + 2 * k; c1 = c0 + 1;
    }
    int ev = 0;
    if (reset == 2) {
        ev = ld_dev(epoch);
        if (ev != EPOCH0 + round && tid == 0) atomicAdd(&cnt->stale_epoch, 1ull);
    }
    if (c0 >= 0) {
        if (tid < 2) {
            const int *f = flags + (tid == 0 ? c0 : c1);
            int spins = 0;
            for (;;) {
                const int v = ld_dev(f);
                if (reset == 2 ? v == ev : v != 0) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 22)) { atomicAdd(&cnt->timeout, 1ull); break; }
            }
        }
        __syncthreads();
        asm volatile("" ::: "memory");
        const int v0 = ld_dev(data + c0 * W + tid), v1 = ld_dev(data + c1 * W + tid);
        if (v0 != round) { if (ld_dev(marker + c0) != round) atomicAdd(&cnt->early, 1ull); else atomicAdd(&cnt->stale_data, 1ull); }
        if (v1 != round) { if (ld_dev(marker + c1) != round) atomicAdd(&cnt->early, 1ull); else atomicAdd(&cnt->stale_data, 1ull); }
        if (tid == 0) atomicAdd(&cnt->handovers, 2ull);
    } else {
        __builtin_amdgcn_s_sleep(40);                     // a leaf's own work
    }
    data[t * W + tid] = round;                            // plain stores from all four waves
    if (wait_stores) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        if (release) asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\tbuffer_wbl2 sc1" ::: "memory");
        __hip_atomic_store(flags + t, reset == 2 ? ev : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(marker + t, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char **argv) {
    const double T = argc > 1 ? atof(argv[1]) : 5.0;
    int *flags, *epoch, *data, *marker, *scratch; Counts *cnt;
    CK(hipMalloc(&flags, NWG * 4)); CK(hipMalloc(&epoch, 4)); CK(hipMalloc(&data, NWG * W * 4)); CK(hipMalloc(&marker, NWG * 4));
    CK(hipMalloc(&scratch, 18 * 256 * 4)); CK(hipMalloc(&cnt, sizeof(Counts)));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int all[][3] = { { 0, 0, 0 }, { 2, 1, 1 }, { 0, 0, 1 }, { 2, 1, 0 }, { 1, 0, 1 } };
    const int ncfg = argc > 2 ? atoi(argv[2]) : 5;
    for (int q = 0; q < ncfg && q < 5; q++) {
        const int *c = all[q];
        const int reset = c[0], wait_stores = c[1], release = c[2];
        const int e0 = EPOCH0;
        CK(hipMemsetAsync(flags, 0, NWG * 4, s)); CK(hipMemsetAsync(data, 0, NWG * W * 4, s)); CK(hipMemsetAsync(marker, 0, NWG * 4, s));
        CK(hipMemsetAsync(cnt, 0, sizeof(Counts), s)); CK(hipMemcpyAsync(epoch, &e0, 4, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
        const double t0 = now_s(); int round = 0;
        while (now_s() - t0 < T) {
            for (int k = 0; k < 2000; k++) {
                round++;
                hipLaunchKernelGGL(k_prepare, dim3(18), dim3(256), 0, s, flags, epoch, scratch, reset, round);
                hipLaunchKernelGGL(k_tree, dim3(NWG), dim3(W), 0, s, flags, epoch, data, marker, reset, wait_stores, release, round, cnt);
            }
            CK(hipStreamSynchronize(s));
        }
        Counts h; CK(hipMemcpy(&h, cnt, sizeof(h), hipMemcpyDeviceToHost));
        printf("reset %d (%s) wait_stores %d release %s: %d rounds in %.1f s, %llu hand-overs: early %llu, stale data %llu, stale epoch %llu, timeouts %llu\n", reset,
               reset == 0 ? "0/1 flags, plain reset" : reset == 1 ? "0/1 flags, device-scope reset" : "round number, never reset", wait_stores,
               release ? "wbl2+wait+store" : "wbl2+store (no wait)", round, now_s() - t0,
               h.handovers, h.early, h.stale_data, h.stale_epoch, h.timeout);
        fflush(stdout);
    }
    return 0;
}
