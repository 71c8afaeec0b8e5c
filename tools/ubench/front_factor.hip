// micro-benchmark: the dense factorisation of ONE LDS-resident front (factor_dense_blk of kernels.hip.h: 16-column pivot chains in
// registers, MFMA updates, look-ahead inside the workgroup) outside the solver -- cycles per phase and per 16-column block, for the
// front shapes on M3500's critical path; variants are compared bitwise with the library's routine and against a host Cholesky.
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ubench/front_factor.hip -o tools/ubench/front_factor
#include "../../aprilsam_amd/csrc/kernels.hip.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace asam;

constexpr int NSTAMP = 64;
#ifdef WITH_R05
// round 5's lds_syrk16 (tile index in a VGPR, K loop not unrolled, four store branches), kept here for the comparison
namespace asam {
template <int NT>
__device__ __forceinline__ void old_lds_syrk16(double *__restrict__ S, int ld, int k_lo, int k_hi, int col_lo, int col_hi, int Rv,
                                           int w0 = 0, int nw = NT / 64) {          // tiles go round-robin over waves [w0, w0 + nw)
    const int wave = (threadIdx.x >> 6) - w0, lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    if (wave < 0 || wave >= nw) return;
    const int ntr = (Rv - col_lo + 15) / 16, ntc = (col_hi - col_lo + 15) / 16;
    const int ntiles = ntc * ntr - ntc * (ntc - 1) / 2;
    const int kw = k_hi - k_lo, nk = (kw + 3) >> 2;
    for (int l = wave; l < ntiles; l += nw) {
        int ti, tj;
        trapezoid_tile(l, ntr, ntc, &ti, &tj);
        const int i0 = col_lo + 16 * ti, j0 = col_lo + 16 * tj;
        const int rj = min(j0 + l15, Rv - 1), ri = min(i0 + l15, Rv - 1);
        d4_t acc = (d4_t){ 0, 0, 0, 0 };
        // C elements of this lane: row i = i0 + l15, columns j = j0 + l4 + 4 * reg
        const int i = i0 + l15;
        int off[4];
#pragma unroll
        for (int reg = 0; reg < 4; reg++) { const int j = j0 + l4 + 4 * reg; off[reg] = (i < Rv && j < col_hi && i >= j) ? j * ld + i : -1; }
        double cv[4];
#pragma unroll
        for (int reg = 0; reg < 4; reg++) cv[reg] = S[off[reg] >= 0 ? off[reg] : 0];
#pragma unroll 4
        for (int ks = 0; ks < nk; ks++) {
            const int kk = 4 * ks + l4;
            const bool kok = kk < kw;
            const double *col = S + (size_t)(k_lo + (kok ? kk : 0)) * ld;
            const double x = col[rj], y = col[ri];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(kok ? x : 0.0, kok ? y : 0.0, acc, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; reg++) if (off[reg] >= 0) S[off[reg]] = cv[reg] - acc[reg];
    }
}

}
#endif

#ifdef WITH_R05
#include "/tmp/old_r05.inc"      // round 5's chain_block / chain_finish / factor_dense_blk under old_ names (git show HEAD~:...; not kept in the tree)
#endif
// VAR 0: the library's factor_dense_blk; VAR 1: the same with per-phase stamps of wave 0 (shader clock); VAR 2 (-DWITH_R05): round 5's routine
template <int NT, int VAR>
__global__ void __launch_bounds__(NT) k_fac(const double *__restrict__ A, double *__restrict__ out, int R, int C, int ns, long long *stamps, int *bad, int inner, int ld) {
    extern __shared__ __attribute__((aligned(16))) double S[];
    const int Rv = R - 2;
    double *stage = S + (size_t)ld * C;
    long long t0 = 0;
    for (int rep = 0; rep < inner; rep++) {          // inner > 1: the last repetition runs with the instruction cache warm
    __syncthreads();
    for (int e = threadIdx.x; e < R * C; e += NT) { const int c = e / R, r = e - c * R; S[(size_t)c * ld + r] = A[e]; }
    __syncthreads();
    t0 = clock64();
    if constexpr (VAR == 0) factor_dense_blk<NT>(S, ld, ns, Rv, C, bad, stage, nullptr);
    else if constexpr (VAR == 2) {
#ifdef WITH_R05
        old_factor_dense_blk<NT>(S, ld, ns, Rv, C, bad, stage, nullptr);
#endif
    } else {
        // stamped copy of factor_dense_blk
        constexpr int NW = NT / 64;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        int si = 0;
        auto st = [&](int) { if (threadIdx.x == 0 && si < NSTAMP) stamps[si++] = clock64() - t0; };
        double Dd[BW];
        auto chain = [&](int k0, int nwc) {
            const int wdt = min(BW, ns - k0);
            if (wdt <= 8) chain_block<NT, 8>(S, ld, k0, wdt, Rv, bad, reinterpret_cast<double (&)[8]>(Dd), nwc); else chain_block<NT, 16>(S, ld, k0, wdt, Rv, bad, Dd, nwc);
        };
        auto finish = [&](int k0) {
            if (wave != 0) return;
            const int wdt = min(BW, ns - k0);
            if (wdt <= 8) chain_store_diag<8>(S, ld, k0, wdt, reinterpret_cast<const double (&)[8]>(Dd)); else chain_store_diag<16>(S, ld, k0, wdt, Dd);
        };
        auto chain_waves = [&](int k0) { const int wdt = min(BW, ns - k0), brows = 64 - (wdt <= 8 ? 8 : 16); return min(max(1, (Rv - (k0 + wdt) + brows - 1) / brows), NW / 2); };
        chain(0, NW); st(0);
        __syncthreads(); st(1);
        finish(0); st(2);
        for (int k0 = 0; k0 < ns; k0 += BW) {
            const int wdt = min(BW, ns - k0), below0 = k0 + wdt;
            if (below0 < ns) {
                const int next_hi = min(below0 + BW, ns), cw = chain_waves(below0);
                lds_syrk16<NT>(S, ld, k0, below0, below0, next_hi, Rv, 1, NW - 1); st(3);
                __syncthreads(); st(4);
                if (wave < cw) chain(below0, cw);
                else if (next_hi < C) lds_syrk16<NT>(S, ld, k0, below0, next_hi, C, Rv, cw, NW - cw);
                st(5);
                __syncthreads(); st(6);
                finish(below0); st(7);
            } else {
                __syncthreads();
                if (below0 < C) lds_syrk16<NT>(S, ld, k0, below0, below0, C, Rv);
                __syncthreads(); st(8);
            }
        }
    }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) stamps[NSTAMP - 1] = t1 - t0;
    for (int e = threadIdx.x; e < R * C; e += NT) { const int c = e / R, r = e - c * R; out[e] = S[(size_t)c * ld + r]; }
}

static void host_factor(std::vector<double> &F, int R, int C, int ns) {      // right-looking, first ns columns; Schur update of the rest (lower part, rows < R - 2)
    const int Rv = R - 2;
    for (int j = 0; j < ns; j++) {
        const double d = std::sqrt(F[(size_t)j * R + j]);
        for (int i = j; i < Rv; i++) F[(size_t)j * R + i] /= d;
        for (int c = j + 1; c < C; c++) { const double l = F[(size_t)j * R + c]; for (int i = c; i < Rv; i++) F[(size_t)c * R + i] -= F[(size_t)j * R + i] * l; }
    }
}

template <int NT, int VAR> static double run(const std::vector<double> &A, std::vector<double> &res, int R, int C, int ns, std::vector<long long> &st, int reps, int inner = 1, int ld = 0) {
    if (!ld) ld = R | 1;
    double *dA, *dO; long long *dS; int *dB;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dO, A.size() * 8); hipMalloc(&dS, NSTAMP * 8); hipMalloc(&dB, 16);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemset(dS, 0, NSTAMP * 8); hipMemset(dB, 0, 16);
    const size_t lds = (size_t)ld * C * 8 + wl_bytes(NT / 64);
    hipFuncSetAttribute((const void *)k_fac<NT, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    long long best = 1ll << 60;
    st.assign(NSTAMP, 0);
    for (int r = 0; r < reps; r++) {
        hipLaunchKernelGGL((k_fac<NT, VAR>), dim3(1), dim3(NT), lds, 0, dA, dO, R, C, ns, dS, dB, inner, ld);
        hipDeviceSynchronize();
        std::vector<long long> s(NSTAMP); hipMemcpy(s.data(), dS, NSTAMP * 8, hipMemcpyDeviceToHost);
        if (s[NSTAMP - 1] < best) { best = s[NSTAMP - 1]; st = s; }
    }
    res.resize(A.size()); hipMemcpy(res.data(), dO, A.size() * 8, hipMemcpyDeviceToHost);
    int b[4]; hipMemcpy(b, dB, 16, hipMemcpyDeviceToHost);
    if (b[0]) printf("  (bad flag %d)\n", b[0]);
    hipFree(dA); hipFree(dO); hipFree(dS); hipFree(dB);
    return (double)best;
}

int main(int argc, char **argv) {
    // (own blocks, update blocks) of fronts on M3500's critical path (profiles/r05_front_times_m3500.txt) + a leaf
    const int shapes[][2] = { { 15, 20 }, { 8, 20 }, { 19, 21 }, { 14, 32 }, { 17, 29 }, { 22, 9 }, { 12, 0 } };
    for (auto &sh : shapes) {
        const int nsb = sh[0], nub = sh[1], nbc = nsb + nub, R = 3 * (nbc + 1), C = 3 * nbc, ns = 3 * nsb;
        if ((size_t)(R | 1) * C * 8 + wl_bytes(16) > 160 * 1024) { printf("nsb %d nub %d: does not fit\n", nsb, nub); continue; }
        std::vector<double> M((size_t)C * C), A((size_t)R * C, 0.0);
        srand(7);
        for (auto &v : M) v = rand() / (double)RAND_MAX - 0.5;
        for (int c = 0; c < C; c++) for (int r = c; r < C; r++) { double a = 0; for (int k = 0; k < C; k++) a += M[(size_t)k * C + r] * M[(size_t)k * C + c]; A[(size_t)c * R + r] = a + (r == c ? C : 0); }
        for (int c = 0; c < C; c++) A[(size_t)c * R + C] = rand() / (double)RAND_MAX;      // right-hand-side row
        std::vector<double> ref = A; host_factor(ref, R, C, ns);
        std::vector<double> r0, r1, r2; std::vector<long long> s0, s1, s2;
        const double c0 = run<1024, 0>(A, r0, R, C, ns, s0, 5), c1 = run<1024, 1>(A, r1, R, C, ns, s1, 5);
        { std::vector<double> rw; std::vector<long long> sw; const double cw = run<1024, 0>(A, rw, R, C, ns, sw, 3, 4); printf("nsb %2d nub %2d: library, 4th repetition inside one launch (warm instruction cache): %6.0f cycles\n", nsb, nub, cw); }
        for (int m : { 15, 17, 16, 1, 9 }) {          // leading dimension of the LDS array: the smallest one >= R that is m (mod 32)
            int ldm = R; while ((ldm & 31) != m) ldm++;
            if ((size_t)ldm * C * 8 + wl_bytes(16) > 160 * 1024) continue;
            std::vector<double> rw; std::vector<long long> sw; const double cw = run<1024, 0>(A, rw, R, C, ns, sw, 5, 1, ldm);
            bool eq = true; for (int c = 0; c < C; c++) for (int r = c; r < R - 2; r++) eq = eq && rw[(size_t)c * R + r] == r0[(size_t)c * R + r];
            printf("   leading dimension %3d (= %2d mod 32; R | 1 = %d): %6.0f cycles, bitwise equal %d\n", ldm, m, R | 1, cw, (int)eq);
        }
#ifdef WITH_R05
        const double c2 = run<1024, 2>(A, r2, R, C, ns, s2, 5);
#else
        const double c2 = 0; r2 = r0; s2.assign(NSTAMP, 0);
#endif
        double err = 0; bool same = true;
        for (int c = 0; c < C; c++) for (int r = c; r < R - 2; r++) { err = fmax(err, fabs(r0[(size_t)c * R + r] - ref[(size_t)c * R + r])); same = same && r0[(size_t)c * R + r] == r1[(size_t)c * R + r] && r0[(size_t)c * R + r] == r2[(size_t)c * R + r]; }
        printf("nsb %2d nub %2d (R %3d, ns %2d): library %6.0f cycles = %.0f per 16 columns | stamped copy %6.0f, round 5's routine %6.0f (0: not compiled in), all bitwise equal %d, max |err| vs host %.2e\n", nsb, nub, R, ns, c0, c0 / ((ns + 15) / 16), c1, c2, (int)same, err);
        printf("   stamps (cycles from start; 0 chain0, 1 barrier, 2 finish | per block: 3 syrk-next, 4 barrier, 5 chain||far, 6 barrier, 7 finish | 8 last update):\n   ");
        for (int i = 0; i < NSTAMP - 1 && s1[i]; i++) printf(" %lld", s1[i]);
        printf("\n");
    }
    return 0;
}
