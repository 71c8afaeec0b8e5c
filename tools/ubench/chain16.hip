// micro-benchmark: the 16-column in-register pivot chain of the small-front kernels (chain_block of kernels.hip.h) on ONE wave -- 64 rows
// of a 16-column block, lanes 0-15 = the diagonal block -- in several instruction schedules; cycles per 16 columns and the results compared.
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ubench/chain16.hip -o tools/ubench/chain16
#include "../../aprilsam_amd/csrc/kernels.hip.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using asam::readlane_d; using asam::fast_rsqrt;
constexpr int BWT = 16;
// V0: the library's schedule (round 2): scale column j; column j+1 first and the next pivot from it; ALL remaining pivot-row scalars by
//     v_readlane; then the eight steps of the next pivot's 1/sqrt, two independent fmas behind each
template <int V> __device__ __forceinline__ void chain(double (&D)[BWT]) {
    if constexpr (V == 0) {
        constexpr int PER = 2;
        double dn = readlane_d(D[0], 0);
        double inv = fast_rsqrt(dn);
#pragma unroll
        for (int j = 0; j < BWT; j++) {
            D[j] *= inv;
            double lc[BWT];
            if (j + 1 < BWT) {
                lc[j + 1] = readlane_d(D[j], j + 1);
                D[j + 1] = fma(-D[j], lc[j + 1], D[j + 1]);
                dn = readlane_d(D[j + 1], j + 1);
            }
#pragma unroll
            for (int c = j + 2; c < BWT; c++) lc[c] = readlane_d(D[j], c);
            __builtin_amdgcn_sched_barrier(0);
            double y = 0, h = 0, e = 0;
#pragma unroll
            for (int st = 0; st < 8; st++) {
                if (j + 1 < BWT) {
                    if (st == 0) { y = __builtin_amdgcn_rsq(dn); h = 0.5 * dn; }
                    else if (st == 1 || st == 4) e = -h * y;
                    else if (st == 2 || st == 5) e = fma(e, y, 0.5);
                    else if (st == 3 || st == 6) y = fma(y, e, y);
                }
#pragma unroll
                for (int q = 0; q < PER; q++) {
                    const int c = j + 2 + PER * st + q;
                    if (c < BWT) D[c] = fma(-D[j], lc[c], D[c]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            inv = y;
        }
    } else if constexpr (V == 1) {
        // V1: the pivot-row scalars of columns >= j + 2 are fetched inside the 1/sqrt window, each right before its fma
        constexpr int PER = 2;
        double dn = readlane_d(D[0], 0);
        double inv = fast_rsqrt(dn);
#pragma unroll
        for (int j = 0; j < BWT; j++) {
            D[j] *= inv;
            if (j + 1 < BWT) {
                const double l1 = readlane_d(D[j], j + 1);
                D[j + 1] = fma(-D[j], l1, D[j + 1]);
                dn = readlane_d(D[j + 1], j + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            double y = 0, h = 0, e = 0;
#pragma unroll
            for (int st = 0; st < 8; st++) {
                if (j + 1 < BWT) {
                    if (st == 0) { y = __builtin_amdgcn_rsq(dn); h = 0.5 * dn; }
                    else if (st == 1 || st == 4) e = -h * y;
                    else if (st == 2 || st == 5) e = fma(e, y, 0.5);
                    else if (st == 3 || st == 6) y = fma(y, e, y);
                }
#pragma unroll
                for (int q = 0; q < PER; q++) {
                    const int c = j + 2 + PER * st + q;
                    if (c < BWT) { const double l = readlane_d(D[j], c); D[c] = fma(-D[j], l, D[c]); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            inv = y;
        }
    } else if constexpr (V == 2) {
        // V2: unscaled form.  Column j is final (all updates applied) long before its pivot's 1/sqrt is: the pivot-row scalars are read from
        // the UNSCALED column as soon as it is final -- inside the previous pivot's 1/sqrt window -- and the update is
        // D[c] -= (D[j] inv^2) * raw_j[c]; the stored column is D[j] inv.  (Rounds differently from V0 in the last bit.)
        double dn = readlane_d(D[0], 0);
        double raw[BWT];                                // raw[c] = unscaled D[j][lane c] of the CURRENT column j
#pragma unroll
        for (int c = 1; c < BWT; c++) raw[c] = readlane_d(D[0], c);
        double inv = fast_rsqrt(dn);
#pragma unroll
        for (int j = 0; j < BWT; j++) {
            const double inv2 = inv * inv;
            const double T = D[j] * inv2;               // column j times 1 / pivot
            D[j] *= inv;                                // the stored column of L
            double nraw[BWT];
            if (j + 1 < BWT) {
                D[j + 1] = fma(-T, raw[j + 1], D[j + 1]);
                dn = readlane_d(D[j + 1], j + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            double y = 0, h = 0, e = 0;
#pragma unroll
            for (int st = 0; st < 8; st++) {
                if (j + 1 < BWT) {
                    if (st == 0) { y = __builtin_amdgcn_rsq(dn); h = 0.5 * dn; }
                    else if (st == 1 || st == 4) e = -h * y;
                    else if (st == 2 || st == 5) e = fma(e, y, 0.5);
                    else if (st == 3 || st == 6) y = fma(y, e, y);
                }
                // this step's remaining updates, two per stage, and -- column j + 1 being final -- its pivot-row scalars for the next step
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int c = j + 2 + 2 * st + q;
                    if (c < BWT) { D[c] = fma(-T, raw[c], D[c]); nraw[c] = readlane_d(D[j + 1], c); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int c = j + 2; c < BWT; c++) raw[c] = nraw[c];
            inv = y;
        }
    }
}
template <int V> __global__ void __launch_bounds__(64) k(const double *in, double *out, long long *cyc, int reps) {
    double D[BWT], acc = 0;
    long long best = 1ll << 60;
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int c = 0; c < BWT; c++) D[c] = in[c * 64 + threadIdx.x];
        __builtin_amdgcn_s_waitcnt(0);
        const long long t0 = clock64();
        chain<V>(D);
        __builtin_amdgcn_sched_barrier(0);
        const long long t1 = clock64();
        if (t1 - t0 < best) best = t1 - t0;
#pragma unroll
        for (int c = 0; c < BWT; c++) acc += D[c];
    }
#pragma unroll
    for (int c = 0; c < BWT; c++) out[c * 64 + threadIdx.x] = D[c];
    out[BWT * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[V] = best;
}
// loads + the bare schedule-0 chain + stores with run-time geometry: what chain_block costs beyond this is its bookkeeping
template <int NT, bool FAST> __device__ __forceinline__ void plain_block(double *S, int ld, int k0, int wdt, int Rv, int nwc, long long *ts) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool isdiag = lane < 16;
    const int below0 = k0 + wdt;
    const int nchunk = max(1, (Rv - below0 + 47) / 48);
    for (int ch = wave; ch < nchunk; ch += nwc) {
        const int row = isdiag ? k0 + lane : below0 + 48 * ch + (lane - 16);
        const bool valid = isdiag ? lane < wdt : row < Rv;
        const int rowc = valid ? row : k0;
        double D[16];
        if (ts && threadIdx.x == 0) ts[0] = clock64();
        if (FAST && wdt == 16) {                      // (wave-uniform) a full block: no redirects
            const double *p = S + k0 * ld + rowc;
#pragma unroll
            for (int c = 0; c < 16; c++) D[c] = p[c * ld];
        } else {
#pragma unroll
            for (int c = 0; c < 16; c++) D[c] = S[(k0 + (c < wdt ? c : 0)) * ld + rowc];
        }
        if (ts) { __builtin_amdgcn_s_waitcnt(0); if (threadIdx.x == 0) ts[1] = clock64(); }
        chain<0>(D);
        if (ts) { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) ts[2] = clock64(); }
        if (valid && !isdiag) {
            if (FAST && wdt == 16) {
                double *q = S + k0 * ld + row;
#pragma unroll
                for (int c = 0; c < 16; c++) q[c * ld] = D[c];
            } else {
#pragma unroll
                for (int c = 15; c >= 0; c--) S[(k0 + (c < wdt ? c : 0)) * ld + row] = (c < wdt) ? D[c] : D[0];
            }
        }
        if (ts) { __builtin_amdgcn_s_waitcnt(0); if (threadIdx.x == 0) ts[3] = clock64(); }
    }
}
template <int NT, bool FAST> __global__ void __launch_bounds__(NT) k_plain(const double *in, long long *cyc, int nwc, int slot, int ld, int Rv, int k0, int wdt, long long *ts) {
    extern __shared__ double S[];
    long long best = 1ll << 60;
    for (int r = 0; r < 20; r++) {
        for (int e = threadIdx.x; e < 16 * 112; e += NT) { const int c = e / 112, rr = e % 112; S[c * ld + rr] = in[c * 64 + (rr < 16 ? rr : 16 + (rr - 16) % 48)]; }
        __syncthreads();
        const long long t0 = clock64();
        if ((int)(threadIdx.x >> 6) < nwc) plain_block<NT, FAST>(S, ld, k0, wdt, Rv, nwc, ts);
        const long long t1 = clock64();
        __syncthreads();
        if (threadIdx.x == 0 && t1 - t0 < best) best = t1 - t0;
    }
    if (threadIdx.x == 0) cyc[slot] = best;
}
// the library's chain_block on an LDS-resident 112 x 16 block (16 + 2 x 48 rows: two chunks), NT threads in the workgroup, `nwc` chain waves
template <int NT> __global__ void __launch_bounds__(NT) k_lib(const double *in, long long *cyc, int nwc, int slot, int ld, int Rv, int k0, int wdt) {
    extern __shared__ double S[];
    long long best = 1ll << 60;
    int bad[4] = { 0, 0, 0, 0 };
    __shared__ int sbad[4];
    for (int r = 0; r < 20; r++) {
        for (int e = threadIdx.x; e < 16 * 112; e += NT) { const int c = e / 112, rr = e % 112; S[c * ld + rr] = in[c * 64 + (rr < 16 ? rr : 16 + (rr - 16) % 48)]; }
        __syncthreads();
        const long long t0 = clock64();
        double Dd[16];
        if ((int)(threadIdx.x >> 6) < nwc) asam::chain_block<NT, 16>(S, ld, k0, wdt, Rv, sbad, Dd, nwc);
        const long long t1 = clock64();
        __syncthreads();
        if (threadIdx.x == 0 && t1 - t0 < best) best = t1 - t0;
    }
    if (threadIdx.x == 0) cyc[slot] = best;
    (void)bad;
}
int main() {
    // a 64 x 16 block whose top 16 x 16 is symmetric positive definite: rows 16.. are rows below the diagonal block
    std::vector<double> A(64 * 16), M(16 * 16);
    srand(3);
    for (auto &v : M) v = rand() / (double)RAND_MAX - 0.5;
    for (int c = 0; c < 16; c++) for (int r = 0; r < 64; r++) {
        if (r < 16) { double a = 0; for (int k = 0; k < 16; k++) a += M[k * 16 + r] * M[k * 16 + c]; A[c * 64 + r] = a + (r == c ? 16 : 0); }
        else A[c * 64 + r] = rand() / (double)RAND_MAX - 0.5;
    }
    std::vector<double> ref = A;                     // host: lower Cholesky of the top block, rows below solved
    for (int j = 0; j < 16; j++) {
        const double d = std::sqrt(ref[j * 64 + j]);
        for (int r = j; r < 64; r++) ref[j * 64 + r] /= d;
        for (int c = j + 1; c < 16; c++) { const double l = ref[j * 64 + c]; for (int r = c; r < 64; r++) if (r >= 16 || r >= c) ref[c * 64 + r] -= ref[j * 64 + r] * l; }
    }
    double *dI, *dO; long long *dC;
    hipMalloc(&dI, A.size() * 8); hipMalloc(&dO, (A.size() + 64) * 8); hipMalloc(&dC, 64);
    hipMemcpy(dI, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemset(dC, 0, 64);
    std::vector<double> o[3];
    for (int v = 0; v < 3; v++) {
        if (v == 0) k<0><<<1, 64>>>(dI, dO, dC, 50); else if (v == 1) k<1><<<1, 64>>>(dI, dO, dC, 50); else k<2><<<1, 64>>>(dI, dO, dC, 50);
        hipDeviceSynchronize();
        o[v].resize(A.size()); hipMemcpy(o[v].data(), dO, A.size() * 8, hipMemcpyDeviceToHost);
    }
    hipLaunchKernelGGL(k_lib<64>, dim3(1), dim3(64), 113 * 16 * 8, 0, dI, dC, 1, 3, 113, 112, 0, 16);
    hipLaunchKernelGGL(k_lib<1024>, dim3(1), dim3(1024), 113 * 16 * 8, 0, dI, dC, 1, 4, 113, 112, 0, 16);
    hipLaunchKernelGGL(k_lib<1024>, dim3(1), dim3(1024), 113 * 16 * 8, 0, dI, dC, 2, 5, 113, 112, 0, 16);
    hipLaunchKernelGGL(k_lib<1024>, dim3(1), dim3(1024), 113 * 16 * 8, 0, dI, dC, 8, 6, 113, 112, 0, 16);
    long long *dT; hipMalloc(&dT, 64); hipMemset(dT, 0, 64);
    for (int fast = 0; fast < 2; fast++) {
        if (fast) hipLaunchKernelGGL((k_plain<1024, true>), dim3(1), dim3(1024), 113 * 16 * 8, 0, dI, dC, 2, 7, 113, 112, 0, 16, (long long *)nullptr);
        else hipLaunchKernelGGL((k_plain<1024, false>), dim3(1), dim3(1024), 113 * 16 * 8, 0, dI, dC, 2, 7, 113, 112, 0, 16, (long long *)nullptr);
        hipDeviceSynchronize();
        { long long h7; hipMemcpy(&h7, dC + 7, 8, hipMemcpyDeviceToHost); printf("loads + bare chain + stores, run-time geometry, 1024 threads, 2 chain waves, full-block fast path %d: %lld cycles\n", fast, h7); }
        if (fast) hipLaunchKernelGGL((k_plain<1024, true>), dim3(1), dim3(1024), 113 * 16 * 8, 0, dI, dC, 2, 7, 113, 112, 0, 16, dT);
        else hipLaunchKernelGGL((k_plain<1024, false>), dim3(1), dim3(1024), 113 * 16 * 8, 0, dI, dC, 2, 7, 113, 112, 0, 16, dT);
        hipDeviceSynchronize();
        { long long t[4]; hipMemcpy(t, dT, 32, hipMemcpyDeviceToHost); printf("   ... stamped: loads %lld, chain %lld, stores %lld cycles\n", t[1] - t[0], t[2] - t[1], t[3] - t[2]); }
    }
    long long h[8]; hipMemcpy(h, dC, 64, hipMemcpyDeviceToHost);
    printf("library chain_block incl. its LDS loads / stores, wave 0's time: 64-thread workgroup, 1 chain wave doing both chunks %lld | 1024 threads: 1 chain wave %lld, 2 chain waves %lld, 8 chain waves (2 busy) %lld cycles\n", h[3], h[4], h[5], h[6]);
    for (int v = 0; v < 3; v++) {
        double err = 0, dv0 = 0;
        for (int c = 0; c < 16; c++) for (int r = c; r < 64; r++) { err = fmax(err, fabs(o[v][c * 64 + r] - ref[c * 64 + r])); dv0 = fmax(dv0, fabs(o[v][c * 64 + r] - o[0][c * 64 + r])); }
        printf("schedule %d: %lld cycles per 16 columns (%.0f per column), max |err| vs host %.2e, max |diff| vs schedule 0 %.2e\n", v, h[v], h[v] / 16.0, err, dv0);
    }
    return 0;
}
