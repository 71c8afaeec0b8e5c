// kernels.hip.h — hand-written HIP kernels (gfx950 / CDNA4, wave64) for the Gauss-Newton hot path.
//
//   k_linearize        xyt / xytpos factor linearisation + 3x3 J^T W J blocks   (april_graph_xyt.c:62-124,
//                      april_graph_xytpos.c:63-102, aprilsam.c:159-192)              HBM-bound streaming
//   k_front_small      one workgroup per small front: gather-assemble in LDS (original blocks + children's
//                      Schur updates), dense right-looking block Cholesky in LDS, write L panel + update
//                      (replaces cs_chol csparse.c:462-512 for these columns)        LDS / latency bound
//   k_assemble_big     chunked gather-assembly of large fronts in HBM
//   k_panel_big        NB-wide panel: diagonal block Cholesky (wave shuffles) + triangular solve of the rows
//   k_syrk_big         trailing update C -= P P^T with v_mfma_f64_16x16x4_f64       FP64-MFMA bound
//   k_backsolve        x_T = L11^-T (y_T - L21^T x_struct), level by level root->leaves (smatd.c:1075)
//   k_update_states    state = l_point + dx, theta wrap, NaN guard (april_graph_xyt.c:302-314)
//   k_chi2 / k_reduce  chi^2 with the 1/2-on-xyt convention (april_graph.c:79-98), deterministic sum
//
// The forward solve U^T y = B (smatd.c:1051) has no kernel of its own: the right-hand side rides along
// as an extra ROW of every front, so the factorisation leaves y in place.
//
// All arithmetic FP64.  No atomics on the data path: every sum has a fixed order => bit-reproducible.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace asam {

constexpr int TPB = 256;          // threads per workgroup (4 waves)
constexpr int NB = 32;            // panel width of the big-front path
constexpr int PANEL_ROWS = 256;   // rows per workgroup in k_panel_big
constexpr int TILE = 64;          // syrk output tile (4 waves x 32x32)
constexpr int ASM_CB = 8;         // block columns per assembly chunk (big fronts)

struct DevPlan {
    int nF;
    const int *f_first, *f_nsb, *f_nub, *f_parent;
    const long long *f_off, *f_rows_ptr;
    const int *f_rows, *f_rel;
    const int *ch_ptr, *ch_idx;
    const int *bd_front_ptr, *bd_row, *bd_col, *bd_src_ptr, *bd_src;
    const int *rd_front_ptr, *rd_col, *rd_src_ptr, *rd_src;
    const double *lambda;         // per elimination position (block): Tikhonov term of its 3 diagonals
};

// ---- work decomposition of the big-front kernels (shared by host launch tables and device decode) ----
__host__ __device__ inline int asm_chunks(int nbc) { return (nbc + ASM_CB - 1) / ASM_CB; }
__host__ __device__ inline int panel_tiles(int R, int ns, int step) {
    int k0 = step * NB, wdt = (ns - k0 < NB) ? ns - k0 : NB;
    int below = (R - 2) - (k0 + wdt);
    int n = (below + PANEL_ROWS - 1) / PANEL_ROWS;
    return n < 1 ? 1 : n;
}
__host__ __device__ inline void syrk_dims(int R, int C, int ns, int step, int *c0, int *ntr, int *ntc) {
    int k0 = step * NB, wdt = (ns - k0 < NB) ? ns - k0 : NB;
    *c0 = k0 + wdt;
    *ntr = ((R - 2) - *c0 + TILE - 1) / TILE;
    *ntc = (C - *c0 + TILE - 1) / TILE;
}
__host__ __device__ inline int syrk_tiles(int R, int C, int ns, int step) {
    int c0, ntr, ntc; syrk_dims(R, C, ns, step, &c0, &ntr, &ntc);
    return ntc * ntr - ntc * (ntc - 1) / 2;
}
// segment a with pre[a] <= bid < pre[a+1]
__device__ __forceinline__ int find_seg(const int *__restrict__ pre, int n, int bid) {
    int lo = 0, hi = n;
    while (hi - lo > 1) { int m = (lo + hi) >> 1; if (pre[m] <= bid) lo = m; else hi = m; }
    return lo;
}

// ------------------------------------------------------------------------------------------------------
// math helpers
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double mod2pi_dev(double v) {   // common/math_util.h:113-122, range [-pi, pi)
    const double TWOPI = 6.2831853071795862319959;
    const double PI_ = 3.141592653589793238462643383279502884196;
    double vin = v + PI_;
    return (vin - TWOPI * floor(vin / TWOPI)) - PI_;
}

// c = a(3x3 row-major)^T * b, accumulating k = 0,1,2 in order like matd_multiply (matd.c:241-247)
__device__ __forceinline__ void at_b(const double *a, const double *b, double *c) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double acc = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) acc += a[k * 3 + i] * b[k * 3 + j];
            c[i * 3 + j] = acc;
        }
}
__device__ __forceinline__ void a_b(const double *a, const double *b, double *c) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double acc = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) acc += a[i * 3 + k] * b[k * 3 + j];
            c[i * 3 + j] = acc;
        }
}
__device__ __forceinline__ void a_v(const double *a, const double *v, double *c) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double acc = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) acc += a[i * 3 + k] * v[k];
        c[i] = acc;
    }
}

// residual + Jacobians of one factor at poses pa (and pb).  b < 0: xytpos prior (J = I).
__device__ __forceinline__ void factor_residual(bool binary, const double *pa, const double *pb, const double *z,
                                                double *J0, double *J1, double *r) {
    if (binary) {
        double xa = pa[0], ya = pa[1], ta = pa[2];
        double xb = pb[0], yb = pb[1], tb = pb[2];
        double sa, ca;
        sincos(ta, &sa, &ca);
        double dx = xb - xa, dy = yb - ya;
        double zh0 = ca * dx + sa * dy, zh1 = -sa * dx + ca * dy, zh2 = tb - ta;
        J0[0] = -ca; J0[1] = -sa; J0[2] = -sa * dx + ca * dy;
        J0[3] = sa;  J0[4] = -ca; J0[5] = -ca * dx - sa * dy;
        J0[6] = 0;   J0[7] = 0;   J0[8] = -1;
        J1[0] = ca;  J1[1] = sa;  J1[2] = 0;
        J1[3] = -sa; J1[4] = ca;  J1[5] = 0;
        J1[6] = 0;   J1[7] = 0;   J1[8] = 1;
        r[0] = z[0] - zh0; r[1] = z[1] - zh1; r[2] = mod2pi_dev(z[2] - zh2);
    } else {
        J0[0] = 1; J0[1] = 0; J0[2] = 0; J0[3] = 0; J0[4] = 1; J0[5] = 0; J0[6] = 0; J0[7] = 0; J0[8] = 1;
        r[0] = z[0] - pa[0]; r[1] = z[1] - pa[1]; r[2] = mod2pi_dev(z[2] - pa[2]);
    }
}
__device__ __forceinline__ double rtWr(const double *w, const double *r) {   // april_graph_xyt.c:112-121
    double X0 = w[0] * r[0] + w[1] * r[1] + w[2] * r[2];
    double X1 = w[3] * r[0] + w[4] * r[1] + w[5] * r[2];
    double X2 = w[6] * r[0] + w[7] * r[1] + w[8] * r[2];
    return r[0] * X0 + r[1] * X1 + r[2] * X2;
}

// ------------------------------------------------------------------------------------------------------
// k_linearize: one thread per factor.
//   Hblk[(3f+0)*9..]  (a,a) block, symmetrised from the reference's upper triangle (aprilsam.c:171)
//   Hblk[(3f+1)*9..]  off-diagonal block in FINAL orientation (rows = the endpoint eliminated later)
//   Hblk[(3f+2)*9..]  (b,b) block;   G[(2f+0)*3..], G[(2f+1)*3..] = J^T W r of a and b
//   all 3x3 blocks row-major [front row offset][front col offset]
// xyt factors linearise at l_point (april_graph_xyt.c:77-78), xytpos at state (april_graph_xytpos.c:83-85).
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) k_linearize(int f_begin, int f_end, const int *__restrict__ fa, const int *__restrict__ fb,
                                                   const double *__restrict__ Z, const double *__restrict__ Wm,
                                                   const double *__restrict__ lp, const double *__restrict__ st,
                                                   const unsigned char *__restrict__ swp, double *__restrict__ Hblk,
                                                   double *__restrict__ G) {
    int f = f_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= f_end) return;
    int a = fa[f], b = fb[f];
    if (a < 0) return;
    double w[9], z[3], J0[9], J1[9], r[3], pa[3], pb[3] = { 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < 9; k++) w[k] = Wm[(size_t)9 * f + k];
#pragma unroll
    for (int k = 0; k < 3; k++) z[k] = Z[(size_t)3 * f + k];
    const bool binary = b >= 0;
    const double *src = binary ? lp : st;
#pragma unroll
    for (int k = 0; k < 3; k++) pa[k] = src[(size_t)3 * a + k];
    if (binary) {
#pragma unroll
        for (int k = 0; k < 3; k++) pb[k] = lp[(size_t)3 * b + k];
    }
    factor_residual(binary, pa, pb, z, J0, J1, r);
    double JtW0[9], H[9], g[3];
    at_b(J0, w, JtW0);                       // J0^T W            (aprilsam.c:162)
    a_b(JtW0, J0, H);                        // (J0^T W) J0       (aprilsam.c:167)
    double *o = Hblk + (size_t)(3 * f) * 9;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) o[i * 3 + j] = (i <= j) ? H[i * 3 + j] : H[j * 3 + i];
    a_v(JtW0, r, g);                         // (J0^T W) r        (aprilsam.c:184-187)
    double *go = G + (size_t)(2 * f) * 3;
    go[0] = g[0]; go[1] = g[1]; go[2] = g[2];
    if (binary) {
        a_b(JtW0, J1, H);                    // (J0^T W) J1: rows a, cols b
        const bool s = swp[f];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) o[9 + i * 3 + j] = s ? H[j * 3 + i] : H[i * 3 + j];
        double JtW1[9];
        at_b(J1, w, JtW1);
        a_b(JtW1, J1, H);
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) o[18 + i * 3 + j] = (i <= j) ? H[i * 3 + j] : H[j * 3 + i];
        a_v(JtW1, r, g);
        go[3] = g[0]; go[4] = g[1]; go[5] = g[2];
    }
}

// per-factor chi^2 at `st` (april_graph.c:79-98: 0.5 r'Wr for xyt via state_eval, r'Wr otherwise)
__global__ void __launch_bounds__(TPB) k_chi2(int F, const int *__restrict__ fa, const int *__restrict__ fb,
                                              const double *__restrict__ Z, const double *__restrict__ Wm,
                                              const double *__restrict__ st, double *__restrict__ out) {
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    int a = fa[f], b = fb[f];
    if (a < 0) { out[f] = 0; return; }
    double w[9], z[3], J0[9], J1[9], r[3], pa[3], pb[3] = { 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < 9; k++) w[k] = Wm[(size_t)9 * f + k];
#pragma unroll
    for (int k = 0; k < 3; k++) { z[k] = Z[(size_t)3 * f + k]; pa[k] = st[(size_t)3 * a + k]; }
    if (b >= 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) pb[k] = st[(size_t)3 * b + k];
    }
    factor_residual(b >= 0, pa, pb, z, J0, J1, r);
    double c = rtWr(w, r);
    out[f] = (b >= 0) ? 0.5 * c : c;
}

// deterministic sum of n doubles into out[0]: ONE workgroup, fixed strided partials + fixed tree
__global__ void __launch_bounds__(1024) k_reduce(int n, const double *__restrict__ in, double *__restrict__ out) {
    __shared__ double s[1024];
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += 1024) acc += in[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int h = 512; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) s[threadIdx.x] += s[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = s[0];
}

// ------------------------------------------------------------------------------------------------------
// gather-assembly of block columns [bc0, bc1) of front t into dst (column-major, leading dimension ld):
//   1. zero (Tikhonov lambda on own diagonals, aprilsam.c:197-204)
//   2. original J^T W J blocks and J^T W r rows, summed per destination in a fixed order
//   3. children's update matrices (extend-add), child after child
// Works on LDS (small fronts, whole front) or HBM (big fronts, one chunk per workgroup).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void assemble_front(const DevPlan &P, double *__restrict__ pool, const double *__restrict__ Hblk,
                                               const double *__restrict__ G, int t, int bc0, int bc1, double *dst, int ld) {
    const int tid = threadIdx.x;
    const int nsb = P.f_nsb[t], nub = P.f_nub[t];
    const int nbc = nsb + nub;          // block columns
    const int R = 3 * (nbc + 1);        // rows incl. rhs block row
    const int rhs_row = 3 * nbc;
    // 1. zero the (block-)lower part of the chunk's columns
    {
        const int r0 = 3 * bc0, nr = R - r0, ncol = 3 * (bc1 - bc0);
        const int first = P.f_first[t];
        for (int e = tid; e < nr * ncol; e += TPB) {
            int c = e / nr, r = e - c * nr;
            int col = 3 * bc0 + c, row = r0 + r;
            double v = 0;
            if (row == col && col < 3 * nsb) v = P.lambda[first + col / 3];
            dst[(size_t)col * ld + row] = v;
        }
    }
    __syncthreads();
    // 2a. 3x3 block destinations with bc0 <= col < bc1 (sorted by (col,row) inside the front)
    {
        int lo = P.bd_front_ptr[t], hi = P.bd_front_ptr[t + 1];
        // first destination with col >= bc0 / col >= bc1 (binary search; every thread redundantly)
        int a = lo, b = hi;
        while (a < b) { int m = (a + b) >> 1; if (P.bd_col[m] < bc0) a = m + 1; else b = m; }
        int d0 = a; b = hi;
        while (a < b) { int m = (a + b) >> 1; if (P.bd_col[m] < bc1) a = m + 1; else b = m; }
        int d1 = a;
        for (int e = tid; e < (d1 - d0) * 9; e += TPB) {
            int d = d0 + e / 9, k = e % 9;
            int i = k / 3, j = k - 3 * i;            // element (row offset i, col offset j)
            int br = P.bd_row[d], bc = P.bd_col[d];
            if (br == bc && i < j) continue;           // strict upper part of a diagonal block: not stored
            double acc = 0;
            for (int s = P.bd_src_ptr[d]; s < P.bd_src_ptr[d + 1]; s++) acc += Hblk[(size_t)P.bd_src[s] * 9 + k];
            dst[(size_t)(3 * bc + j) * ld + 3 * br + i] += acc;
        }
        lo = P.rd_front_ptr[t]; hi = P.rd_front_ptr[t + 1];
        a = lo; b = hi;
        while (a < b) { int m = (a + b) >> 1; if (P.rd_col[m] < bc0) a = m + 1; else b = m; }
        d0 = a; b = hi;
        while (a < b) { int m = (a + b) >> 1; if (P.rd_col[m] < bc1) a = m + 1; else b = m; }
        d1 = a;
        for (int e = tid; e < (d1 - d0) * 3; e += TPB) {
            int d = d0 + e / 3, j = e % 3;
            double acc = 0;
            for (int s = P.rd_src_ptr[d]; s < P.rd_src_ptr[d + 1]; s++) acc += G[(size_t)P.rd_src[s] * 3 + j];
            dst[(size_t)(3 * P.rd_col[d] + j) * ld + rhs_row] += acc;
        }
    }
    __syncthreads();
    // 3. children, one after the other (fixed order => deterministic)
    const int wave = tid >> 6, lane = tid & 63;
    for (int ci = P.ch_ptr[t]; ci < P.ch_ptr[t + 1]; ci++) {
        const int c = P.ch_idx[ci];
        const int cns = P.f_nsb[c], cnu = P.f_nub[c];
        const int cR = 3 * (cns + cnu + 1);
        const double *U = pool + P.f_off[c];
        const int *rel = P.f_rel + P.f_rows_ptr[c];
        // child struct blocks jb with bc0 <= rel[jb] < bc1 (rel is strictly increasing)
        int a = 0, b = cnu;
        while (a < b) { int m = (a + b) >> 1; if (rel[m] < bc0) a = m + 1; else b = m; }
        int j0 = a; b = cnu;
        while (a < b) { int m = (a + b) >> 1; if (rel[m] < bc1) a = m + 1; else b = m; }
        int j1 = a;
        // scalar child columns 3*(cns+jb)+j, distributed over waves; lanes run down the rows
        for (int cc = 3 * j0 + wave; cc < 3 * j1; cc += TPB / 64) {
            int jb = cc / 3, j = cc - 3 * jb;
            int ccol = 3 * (cns + jb) + j;
            int pcol = 3 * rel[jb] + j;
            const double *ucol = U + (size_t)ccol * cR;
            double *dcol = dst + (size_t)pcol * ld;
            // rows: from the diagonal element down to the rhs row (inclusive); pad rows skipped
            for (int rr = ccol + lane; rr <= cR - 3; rr += 64) {
                int ib = rr / 3 - cns, i = rr % 3;
                int prow = (ib < cnu) ? 3 * rel[ib] + i : rhs_row;
                dcol[prow] += ucol[rr];
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------
// dense right-looking Cholesky of the first nsb block columns of an LDS-resident front, rank-3 steps.
// S: column-major, leading dimension ld, nbr = nbc+1 block rows (last = rhs row + 2 zero pad rows).
// Leaves L (incl. the solved rhs row y) in the first 3*nsb columns and the Schur update in the rest.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void factor_front_lds(double *S, int ld, int nsb, int nbc, int *bad) {
    const int tid = threadIdx.x;
    const int nbr = nbc + 1, R = 3 * nbr;
    const int tx = tid & 15, ty = tid >> 4;      // 16 x 16 thread grid over (block row, block col) tiles
    for (int kb = 0; kb < nsb; kb++) {
        const int k = 3 * kb;
        const double *Dk = S + (size_t)k * ld + k;
        double d00 = Dk[0], d10 = Dk[1], d20 = Dk[2], d11 = Dk[ld + 1], d21 = Dk[ld + 2], d22 = Dk[2 * ld + 2];
        double l00 = sqrt(d00), i00 = 1.0 / l00;
        double l10 = d10 * i00, l20 = d20 * i00;
        double t11 = d11 - l10 * l10;
        double l11 = sqrt(t11), i11 = 1.0 / l11;
        double l21 = (d21 - l20 * l10) * i11;
        double t22 = d22 - l20 * l20 - l21 * l21;
        double l22 = sqrt(t22), i22 = 1.0 / l22;
        if (tid == 0 && !(d00 > 0 && t11 > 0 && t22 > 0)) *bad = 1;
        // panel: rows below the diagonal block
        for (int i = k + 3 + tid; i < R; i += TPB) {
            double x0 = S[(size_t)k * ld + i], x1 = S[(size_t)(k + 1) * ld + i], x2 = S[(size_t)(k + 2) * ld + i];
            double y0 = x0 * i00;
            double y1 = (x1 - y0 * l10) * i11;
            double y2 = (x2 - y0 * l20 - y1 * l21) * i22;
            S[(size_t)k * ld + i] = y0; S[(size_t)(k + 1) * ld + i] = y1; S[(size_t)(k + 2) * ld + i] = y2;
        }
        __syncthreads();
        if (tid == 0) {   // the diagonal block itself is not read by the trailing update
            double *Dw = S + (size_t)k * ld + k;
            Dw[0] = l00; Dw[1] = l10; Dw[2] = l20; Dw[ld + 1] = l11; Dw[ld + 2] = l21; Dw[2 * ld + 2] = l22;
            Dw[ld] = 0; Dw[2 * ld] = 0; Dw[2 * ld + 1] = 0;
        }
        // trailing update: block tiles (bi >= bj > kb)
        for (int bj = kb + 1 + ty; bj < nbc; bj += 16) {
            double yj[9];
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
                for (int a = 0; a < 3; a++) yj[a * 3 + q] = S[(size_t)(k + q) * ld + 3 * bj + a];
            for (int bi = bj + tx; bi < nbr; bi += 16) {
                double yi[9];
#pragma unroll
                for (int q = 0; q < 3; q++)
#pragma unroll
                    for (int a = 0; a < 3; a++) yi[a * 3 + q] = S[(size_t)(k + q) * ld + 3 * bi + a];
#pragma unroll
                for (int c = 0; c < 3; c++)
#pragma unroll
                    for (int a = 0; a < 3; a++) {
                        double *p = S + (size_t)(3 * bj + c) * ld + 3 * bi + a;
                        double v = *p;
                        v = fma(-yi[a * 3 + 0], yj[c * 3 + 0], v);
                        v = fma(-yi[a * 3 + 1], yj[c * 3 + 1], v);
                        v = fma(-yi[a * 3 + 2], yj[c * 3 + 2], v);
                        *p = v;
                    }
            }
        }
        __syncthreads();
    }
}

// one workgroup per small front of a level; fronts[] lists them.  Dynamic LDS: (R|1) * C doubles.
__global__ void __launch_bounds__(TPB) k_front_small(DevPlan P, const int *__restrict__ fronts, double *__restrict__ pool,
                                                     const double *__restrict__ Hblk, const double *__restrict__ G, int *bad) {
    extern __shared__ __attribute__((aligned(16))) double S[];
    const int t = fronts[blockIdx.x];
    const int nsb = P.f_nsb[t], nub = P.f_nub[t], nbc = nsb + nub;
    const int R = 3 * (nbc + 1), C = 3 * nbc, ld = R | 1;
    assemble_front(P, pool, Hblk, G, t, 0, nbc, S, ld);
    factor_front_lds(S, ld, nsb, nbc, bad);
    // store the block-lower trapezoid back to HBM (L panel + update block)
    double *Fg = pool + P.f_off[t];
    for (int c = threadIdx.x >> 6; c < C; c += TPB / 64) {
        int r0 = 3 * (c / 3);
        for (int r = r0 + (threadIdx.x & 63); r < R; r += 64) Fg[(size_t)c * R + r] = S[(size_t)c * ld + r];
    }
}

// ------------------------------------------------------------------------------------------------------
// big fronts
// ------------------------------------------------------------------------------------------------------
// work item: one chunk of ASM_CB block columns of one big front; list/pre = launch table of the level
__global__ void __launch_bounds__(TPB) k_assemble_big(DevPlan P, const int *__restrict__ list, const int *__restrict__ pre, int n,
                                                      double *__restrict__ pool, const double *__restrict__ Hblk,
                                                      const double *__restrict__ G) {
    const int a = find_seg(pre, n, blockIdx.x);
    const int t = list[a], bc0 = (blockIdx.x - pre[a]) * ASM_CB;
    const int nbc = P.f_nsb[t] + P.f_nub[t];
    const int bc1 = min(bc0 + ASM_CB, nbc);
    assemble_front(P, pool, Hblk, G, t, bc0, bc1, pool + P.f_off[t], 3 * (nbc + 1));
}

// f64 wave shuffle
__device__ __forceinline__ double shfl_d(double v, int src) {
    int lo = __shfl(__double2loint(v), src, 64), hi = __shfl(__double2hiint(v), src, 64);
    return __hiloint2double(hi, lo);
}

// panel step `step` of the big fronts listed in w_front (one workgroup per (front, 256-row tile)):
// factor the NB x NB diagonal block (every workgroup redundantly, wave 0, register-resident rows and
// wave shuffles: no barriers), then solve its rows against it.
__global__ void __launch_bounds__(TPB) k_panel_big(DevPlan P, const int *__restrict__ list, const int *__restrict__ pre, int n,
                                                   int step, double *__restrict__ pool, int *bad) {
    __shared__ double Ld[NB][NB + 1];     // factored diagonal block, lower; Ld[c][c] holds 1/L[c][c]
    const int seg = find_seg(pre, n, blockIdx.x);
    const int t = list[seg], tile = blockIdx.x - pre[seg];
    const int nsb = P.f_nsb[t], nbc = nsb + P.f_nub[t];
    const int R = 3 * (nbc + 1), ns = 3 * nsb;
    const int k0 = step * NB, wdt = min(NB, ns - k0);
    double *Fg = pool + P.f_off[t];
    const int tid = threadIdx.x;
    if (tid < 64) {
        const int r = tid & 31;           // lanes 32..63 mirror lanes 0..31 (keeps shuffles in range)
        double D[NB];
#pragma unroll
        for (int c = 0; c < NB; c++) {
            double v = (r == c) ? 1.0 : 0.0;
            if (r < wdt && c < wdt && c <= r) v = Fg[(size_t)(k0 + c) * R + k0 + r];
            D[c] = v;
        }
        int isbad = 0;
#pragma unroll
        for (int j = 0; j < NB; j++) {
            double djj = shfl_d(D[j], j);
            if (!(djj > 0)) isbad = 1;
            double dj = sqrt(djj), inv = 1.0 / dj;
            D[j] = (r == j) ? dj : ((r > j) ? D[j] * inv : D[j]);
#pragma unroll
            for (int c = j + 1; c < NB; c++) {
                double lcj = shfl_d(D[j], c);
                if (r >= c) D[c] = fma(-D[j], lcj, D[c]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (tid < 32) {
#pragma unroll
            for (int c = 0; c < NB; c++) Ld[r][c] = (c < r) ? D[c] : ((c == r) ? 1.0 / D[c] : 0.0);
            if (tile == 0 && r < wdt) {
#pragma unroll
                for (int c = 0; c < NB; c++) if (c <= r && c < wdt) Fg[(size_t)(k0 + c) * R + k0 + r] = D[c];
            }
            if (isbad && r == 0) *bad = 1;
        }
    }
    __syncthreads();
    // rows below the diagonal block: row = k0 + wdt + tile*PANEL_ROWS + tid.  Column sweep of
    // y L^T = x with the row in registers; L comes from LDS as wave-wide broadcasts.
    const int row = k0 + wdt + tile * PANEL_ROWS + tid;
    if (row <= R - 3) {
        // loads/stores of the columns beyond a partial panel (c >= wdt) are redirected to column 0 with
        // selects instead of branches: hipcc turns 32 conditional stores into a register blow-up + spills
        double y[NB];
#pragma unroll
        for (int c = 0; c < NB; c++) {
            double v = Fg[(size_t)(k0 + (c < wdt ? c : 0)) * R + row];
            y[c] = (c < wdt) ? v : 0.0;
        }
#pragma unroll
        for (int c = 0; c < NB; c++) {
            y[c] *= Ld[c][c];
#pragma unroll
            for (int p = 0; p < NB; p++) if (p > c) y[p] = fma(-y[c], Ld[p][c], y[p]);
        }
#pragma unroll
        for (int c = NB - 1; c >= 0; c--) Fg[(size_t)(k0 + (c < wdt ? c : 0)) * R + row] = (c < wdt) ? y[c] : y[0];
    }
}

// trailing update of panel step `step`: C[i,j] -= sum_p P[i,p] P[j,p] for j >= k0+wdt, i >= j.
// One workgroup per 64x64 tile (w_ti >= w_tj, in units of TILE from c0 = k0 + wdt); 4 waves, each a 32x32
// quadrant as 2x2 v_mfma_f64_16x16x4_f64 tiles.  The MFMA computes the TRANSPOSED update (A = P_j rows,
// B = P_i rows) so that a lane's 4 results sit in consecutive... columns of one row run: stores coalesce
// along rows i (lanes 0-15 = 16 consecutive rows of one column).
typedef double d4_t __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(TPB) k_syrk_big(DevPlan P, const int *__restrict__ list, const int *__restrict__ pre, int n,
                                                  int step, double *__restrict__ pool) {
    const int seg = find_seg(pre, n, blockIdx.x);
    const int t = list[seg];
    const int nsb = P.f_nsb[t], nbc = nsb + P.f_nub[t];
    const int R = 3 * (nbc + 1), C = 3 * nbc, ns = 3 * nsb;
    const int k0 = step * NB, wdt = min(NB, ns - k0);
    int c0, ntr, ntc;
    syrk_dims(R, C, ns, step, &c0, &ntr, &ntc);
    // tile l of the column-major lower trapezoid: columns tj hold (ntr - tj) tiles, tj < ntc
    int l = blockIdx.x - pre[seg];
    int tj = (int)(((2.0 * ntr + 1.0) - sqrt((2.0 * ntr + 1.0) * (2.0 * ntr + 1.0) - 8.0 * l)) * 0.5);
    if (tj < 0) tj = 0;
    if (tj > ntc - 1) tj = ntc - 1;
    while (tj > 0 && tj * ntr - tj * (tj - 1) / 2 > l) tj--;
    while (tj + 1 < ntc && (tj + 1) * ntr - (tj + 1) * tj / 2 <= l) tj++;
    const int ti = tj + (l - (tj * ntr - tj * (tj - 1) / 2));
    const int Rv = R - 2;                             // valid rows (rhs row included, pad rows excluded)
    double *Fg = pool + P.f_off[t];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i0 = c0 + ti * TILE + (wave & 1) * 32;    // row origin of this wave's quadrant
    const int j0 = c0 + tj * TILE + (wave >> 1) * 32;   // col origin
    if (i0 >= Rv || j0 >= C) return;
    if (i0 + 31 < j0) return;                         // quadrant strictly above the diagonal
    const int l15 = lane & 15, l4 = lane >> 4;
    d4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = (d4_t){ 0, 0, 0, 0 };
    const int nk = (wdt + 3) >> 2;
    for (int ks = 0; ks < nk; ks++) {
        const int kk = k0 + 4 * ks + l4;               // panel column this lane supplies
        const bool kok = (4 * ks + l4) < wdt;
        double pj[2], pi[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            int rj = j0 + 16 * q + l15, ri = i0 + 16 * q + l15;
            pj[q] = (kok && rj < C) ? Fg[(size_t)kk * R + rj] : 0.0;      // rows j of the panel (j < C <= Rv)
            pi[q] = (kok && ri < Rv) ? Fg[(size_t)kk * R + ri] : 0.0;
        }
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
                acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(pj[a], pi[b], acc[a][b], 0, 0, 0);
    }
    // D[a][b] element (row = l4 + 4*reg, col = l15) = update of C[i = i0+16b+l15, j = j0+16a+l4+4*reg]
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                int i = i0 + 16 * b + l15, j = j0 + 16 * a + l4 + 4 * reg;
                if (i < Rv && j < C && i >= j) {
                    double *p = Fg + (size_t)j * R + i;
                    *p = *p - acc[a][b][reg];
                }
            }
}

// ------------------------------------------------------------------------------------------------------
// backward substitution, one workgroup per front, levels from the root down.
//   x_T = L11^-T ( y_T - L21^T x_struct )      (row-dot form of smatd_utriangle_solve, smatd.c:1075)
// xw (LDS, R doubles) holds x over the front's rows: struct part gathered from the global x, own part
// filled as it is solved, NB columns at a time from the last block to the first.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {     // result valid in lane 0 (fixed tree order)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        int lo = __shfl_down(__double2loint(v), off, 64), hi = __shfl_down(__double2hiint(v), off, 64);
        v += __hiloint2double(hi, lo);
    }
    return v;
}

__global__ void __launch_bounds__(TPB) k_backsolve(DevPlan P, const int *__restrict__ fronts, const double *__restrict__ pool,
                                                   double *__restrict__ x) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int t = fronts[blockIdx.x];
    const int nsb = P.f_nsb[t], nub = P.f_nub[t], nbc = nsb + nub;
    const int R = 3 * (nbc + 1), ns = 3 * nsb, m = 3 * nbc;
    double *xw = smem;                 // m doubles: x over the front's rows
    double *part = smem + m;           // NB doubles: right-hand side of the current column block
    const double *Fg = pool + P.f_off[t];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int *rows = P.f_rows + P.f_rows_ptr[t];
    for (int e = tid; e < 3 * nub; e += TPB) xw[ns + e] = x[(size_t)3 * rows[e / 3] + e % 3];
    __syncthreads();
    for (int k1 = ns; k1 > 0; k1 -= NB) {
        const int k0 = max(0, k1 - NB), wdt = k1 - k0;
        // w_c = y_c - sum_{i >= k1} L[i,c] x[i] for the wdt columns: wave per column, lanes over rows
        for (int c = wave; c < wdt; c += TPB / 64) {
            const double *col = Fg + (size_t)(k0 + c) * R;
            double acc = 0;
            for (int i = k1 + lane; i < m; i += 64) acc = fma(col[i], xw[i], acc);
            acc = wave_sum(acc);
            if (lane == 0) part[c] = col[m] - acc;          // row m = solved rhs row y
        }
        __syncthreads();
        // in-block solve L[k0..k1)^T x = w on wave 0: lane c keeps column c of the block in registers,
        // x_i is broadcast by shuffle, no LDS traffic and no barriers inside the 32-step recurrence
        if (wave == 0) {
            const int c = lane & 31;
            double Lc[NB];
#pragma unroll
            for (int i = 0; i < NB; i++) Lc[i] = (c < wdt && i < wdt && i >= c) ? Fg[(size_t)(k0 + c) * R + k0 + i] : ((i == c) ? 1.0 : 0.0);
            double w = (c < wdt) ? part[c] : 0.0;
#pragma unroll
            for (int i = NB - 1; i >= 0; i--) {
                double xi = shfl_d((c == i) ? w / Lc[i] : 0.0, i);   // lane i holds L[i][i] and the finished w_i
                if (c == i) w = xi;
                else if (c < i) w = fma(-Lc[i], xi, w);
            }
            if (lane < wdt) xw[k0 + lane] = w;
        }
        __syncthreads();
    }
    const int first = P.f_first[t];
    for (int e = tid; e < ns; e += TPB) x[(size_t)3 * first + e] = xw[e];
}

// state = l_point + dx with theta wrap; NaN in dx leaves the node untouched (april_graph_xyt.c:302-314)
__global__ void __launch_bounds__(TPB) k_update_states(int N, const int *__restrict__ pos, const double *__restrict__ x,
                                                       const double *__restrict__ lp, double *__restrict__ st,
                                                       double *__restrict__ dX) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double *d = x + (size_t)3 * pos[i];
    double d0 = d[0], d1 = d[1], d2 = d[2];
    if (isnan(d0) || isnan(d1) || isnan(d2)) return;
    st[3 * i + 0] = lp[3 * i + 0] + d0;
    st[3 * i + 1] = lp[3 * i + 1] + d1;
    st[3 * i + 2] = mod2pi_dev(lp[3 * i + 2] + d2);
    dX[3 * i + 0] = d0; dX[3 * i + 1] = d1; dX[3 * i + 2] = d2;
}

}  // namespace asam
