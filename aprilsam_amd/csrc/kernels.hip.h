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

// Device-side plan.  Everything a workgroup needs to know about its front comes from ONE 64-byte
// descriptor (a single scalar load) instead of a chain of dependent index lookups: on M3500-sized
// problems the kernels are latency bound and every dependent global load costs ~0.8 us.
struct FrontDesc {
    long long off;                 // offset (doubles) of the frontal array in the pool
    int nsb, nub, first;           // own blocks, update blocks, first own elimination position
    int dest_begin, dest_end;      // DestRec range (sorted by block col, then block row; rhs row = block row nbc)
    int ch_begin, ch_end;          // ChildRec range
    int rows_begin;                // first entry of this front's struct rows in f_rows
    int parent;
    int pad[5];
};
static_assert(sizeof(FrontDesc) == 64, "FrontDesc must stay one cache line");
struct DestRec { int brow, bcol, src_begin, src_end; };     // contributions = slots [src_begin, src_end)
struct ChildRec {
    long long uoff;                // pool offset of the child's update block origin (row = col = 3*cns)
    int cR;                        // leading dimension of the child's frontal array
    int cnu;                       // child's update blocks
    int rel_begin;                 // first entry of the child's block map in f_rel
    int pad;
};
static_assert(sizeof(ChildRec) == 24, "ChildRec layout");

struct DevPlan {
    int nF;
    const FrontDesc *fd;
    const DestRec *dest;
    const ChildRec *child;
    const int *f_rows, *f_rel;
    const int *slot_blk, *slot_rhs;   // per factor: where k_linearize stores its 3 blocks / 2 rhs segments
    const double *lambda;         // per elimination position (block): Tikhonov term of its 3 diagonals
    long long *prof;              // debug: 8 wall-clock stamps (100 MHz) per front, or null
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

// 1/sqrt(d) from v_rsq_f64 + two Newton steps (full double precision for normal positive d; the
// compiler's sqrt()/division expand to ~25 dependent FP64 ops with denormal scaling and fix-ups, which
// sits squarely on the per-column critical path of every Cholesky kernel below)
__device__ __forceinline__ double fast_rsqrt(double d) {
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    double e = fma(-h * y, y, 0.5);
    y = fma(y, e, y);
    e = fma(-h * y, y, 0.5);
    y = fma(y, e, y);
    return y;
}
// broadcast of lane `lane` (wave-uniform index) through SGPRs: v_readlane_b32, no LDS round trip
__device__ __forceinline__ double readlane_d(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
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
//   slot_blk[3f+0]  (a,a) block, symmetrised from the reference's upper triangle (aprilsam.c:171)
//   slot_blk[3f+1]  off-diagonal block in FINAL orientation (rows = the endpoint eliminated later)
//   slot_blk[3f+2]  (b,b) block;   slot_rhs[2f+0], slot_rhs[2f+1] = J^T W r of a and b (first 3 doubles)
//   every slot is 9 doubles of Hc, blocks row-major [front row offset][front col offset]; slots are numbered
//   in the order the assembling front consumes them
// xyt factors linearise at l_point (april_graph_xyt.c:77-78), xytpos at state (april_graph_xytpos.c:83-85).
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) k_linearize(int f_begin, int f_end, const int *__restrict__ fa, const int *__restrict__ fb,
                                                   const double *__restrict__ Z, const double *__restrict__ Wm,
                                                   const double *__restrict__ lp, const double *__restrict__ st,
                                                   const unsigned char *__restrict__ swp, const int *__restrict__ slot_blk,
                                                   const int *__restrict__ slot_rhs, double *__restrict__ Hc) {
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
    double *o = Hc + (size_t)slot_blk[3 * f] * 9;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) o[i * 3 + j] = (i <= j) ? H[i * 3 + j] : H[j * 3 + i];
    a_v(JtW0, r, g);                         // (J0^T W) r        (aprilsam.c:184-187)
    double *go = Hc + (size_t)slot_rhs[2 * f] * 9;
    go[0] = g[0]; go[1] = g[1]; go[2] = g[2];
    if (binary) {
        a_b(JtW0, J1, H);                    // (J0^T W) J1: rows a, cols b
        const bool s = swp[f];
        double *o1 = Hc + (size_t)slot_blk[3 * f + 1] * 9, *o2 = Hc + (size_t)slot_blk[3 * f + 2] * 9;
        double *g1 = Hc + (size_t)slot_rhs[2 * f + 1] * 9;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) o1[i * 3 + j] = s ? H[j * 3 + i] : H[i * 3 + j];
        double JtW1[9];
        at_b(J1, w, JtW1);
        a_b(JtW1, J1, H);
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) o2[i * 3 + j] = (i <= j) ? H[i * 3 + j] : H[j * 3 + i];
        a_v(JtW1, r, g);
        g1[0] = g[0]; g1[1] = g[1]; g1[2] = g[2];
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
//   3. children's update matrices (extend-add), child after child (fixed order => deterministic)
// Works on LDS (small fronts, whole front) or HBM/L2 (medium fronts whole, big fronts one chunk per
// workgroup).  NT = threads of the workgroup.  scratch: LDS ints, >= MAXC * (rows of the front) + 64.
// ------------------------------------------------------------------------------------------------------
constexpr int MAXC = 8;     // children whose metadata / row maps are staged together
constexpr int EPT = 8;      // update elements each thread keeps in flight per pipelined round

struct ChildDesc {          // one per staged child, lives in LDS
    const double *U;        // origin of the child's update block inside its frontal array
    int cR;                 // leading dimension of the child's frontal array
    int jc0, ncol, nr;      // scalar update columns [jc0, jc0+ncol) land in this chunk; rows [jc0, jc0+nr)
    int total;              // nr * ncol
    int pm;                 // offset of this child's row map in the scratch
    int nrow;               // 3*cnu + 1 update rows incl. the rhs row
    const int *rel;         // the child's struct-block -> parent-block map
};

template <int NT>
__device__ __forceinline__ void assemble_front(const DevPlan &P, const FrontDesc &D, const double *__restrict__ pool,
                                               const double *__restrict__ Hc, int bc0, int bc1, bool full,
                                               double *__restrict__ dst, int ld, int *__restrict__ scratch, ChildDesc *cd) {
    const int tid = threadIdx.x;
    const int nsb = D.nsb, nbc = D.nsb + D.nub;     // own / all block columns
    const int R = 3 * (nbc + 1);                    // rows incl. rhs block row
    const int rhs_row = 3 * nbc;
    // 1. zero the (block-)lower part of the chunk's columns
    {
        const int r0 = 3 * bc0, nr = R - r0, ncol = 3 * (bc1 - bc0);
        for (int e = tid; e < nr * ncol; e += NT) {
            int c = e / nr, r = e - c * nr;
            int col = 3 * bc0 + c, row = r0 + r;
            double v = 0;
            if (row == col && col < 3 * nsb) v = P.lambda[D.first + col / 3];
            dst[(size_t)col * ld + row] = v;
        }
    }
    __syncthreads();
    // 2. destination blocks with bc0 <= bcol < bc1 (3x3 J^T W J blocks and 1x3 J^T W r rows alike): one record
    //    load, then the contributions of a destination are consecutive slots of Hc
    {
        int d0 = D.dest_begin, d1 = D.dest_end;
        if (!full) {
            int a = d0, b = d1;
            while (a < b) { int m = (a + b) >> 1; if (P.dest[m].bcol < bc0) a = m + 1; else b = m; }
            d0 = a; b = d1;
            while (a < b) { int m = (a + b) >> 1; if (P.dest[m].bcol < bc1) a = m + 1; else b = m; }
            d1 = a;
        }
        for (int e = tid; e < (d1 - d0) * 9; e += NT) {
            const int d = d0 + e / 9, k = e % 9;
            const int i = k / 3, j = k - 3 * i;            // element (row offset i, col offset j)
            const DestRec rec = P.dest[d];
            const bool rhs = rec.brow == nbc;
            if (rhs ? (i > 0) : (rec.brow == rec.bcol && i < j)) continue;   // rhs: one row; diagonal block: lower part
            double acc = 0;
            for (int q = rec.src_begin; q < rec.src_end; q++) acc += Hc[(size_t)q * 9 + k];   // rhs slots keep g in [0..2]: k = j
            dst[(size_t)(3 * rec.bcol + j) * ld + 3 * rec.brow + i] += acc;
        }
    }
    __syncthreads();
    // 3. children in batches of MAXC: descriptors, then all row maps of the batch in one sweep, then a
    //    two-stage software pipeline over (child, round): the loads of round k+1 are issued before the
    //    read-modify-write of round k, so the memory latency of every child but the first hides behind it.
    for (int cb = D.ch_begin; cb < D.ch_end; cb += MAXC) {
        const int nb = min(MAXC, D.ch_end - cb);
        if (tid < nb) {
            const ChildRec cr = P.child[cb + tid];
            const int *__restrict__ rel = P.f_rel + cr.rel_begin;
            int j0 = 0, j1 = cr.cnu;
            if (!full) {      // child struct blocks jb with bc0 <= rel[jb] < bc1 (rel is strictly increasing)
                int a = 0, b = cr.cnu;
                while (a < b) { int m = (a + b) >> 1; if (rel[m] < bc0) a = m + 1; else b = m; }
                j0 = a; b = cr.cnu;
                while (a < b) { int m = (a + b) >> 1; if (rel[m] < bc1) a = m + 1; else b = m; }
                j1 = a;
            }
            ChildDesc d;
            d.U = pool + cr.uoff;
            d.cR = cr.cR; d.nrow = 3 * cr.cnu + 1;
            d.jc0 = 3 * j0; d.ncol = 3 * (j1 - j0); d.nr = d.nrow - d.jc0; d.total = d.nr * d.ncol;
            d.pm = tid * R; d.rel = rel;
            cd[tid] = d;
        }
        __syncthreads();
        for (int e = tid; e < nb * R; e += NT) {       // (child q, update row i) -> row of this front
            const int q = e / R, i = e - q * R;
            const int nrow = cd[q].nrow;
            if (i < nrow) scratch[e] = (i < nrow - 1) ? 3 * cd[q].rel[i / 3] + (i % 3) : rhs_row;
        }
        __syncthreads();
        double vA[EPT], vB[EPT]; int aA[EPT], aB[EPT];
        auto fetch = [&](int q, int rr, double *v, int *da) {
            const ChildDesc d = cd[q];
#pragma unroll
            for (int k = 0; k < EPT; k++) {
                const int e = (rr * EPT + k) * NT + tid;
                const int cc = (e < d.total) ? e / d.nr : 0, r = (e < d.total) ? e - cc * d.nr : 0;
                const bool ok = e < d.total && r >= cc;
                v[k] = ok ? d.U[(size_t)(d.jc0 + cc) * d.cR + d.jc0 + r] : 0.0;
                da[k] = ok ? scratch[d.pm + d.jc0 + cc] * ld + scratch[d.pm + d.jc0 + r] : -1;
            }
        };
        auto apply = [&](const double *v, const int *da) {
#pragma unroll
            for (int k = 0; k < EPT; k++) if (da[k] >= 0) dst[da[k]] += v[k];
        };
        auto advance = [&](int &q, int &rr) -> bool {      // next (child, round); false when exhausted
            rr++;
            while (q < nb && rr * EPT * NT >= cd[q].total) { q++; rr = 0; }
            return q < nb;
        };
        int q = 0, rr = -1;
        bool more = advance(q, rr);
        if (more) fetch(q, rr, vA, aA);
        while (more) {
            int q2 = q, rr2 = rr;
            const bool more2 = advance(q2, rr2);
            if (more2) fetch(q2, rr2, vB, aB);
            apply(vA, aA);
            __syncthreads();
            if (!more2) break;
            q = q2; rr = rr2;
            more = advance(q, rr);
            if (more) fetch(q, rr, vA, aA);
            apply(vB, aB);
            __syncthreads();
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------
// blocked right-looking Cholesky of the first ns columns of a front, NB columns per panel:
//   (a) wave 0 factors the NB x NB diagonal block with its rows in registers, broadcasting through
//       v_readlane (no LDS, no barrier inside the 32-step recurrence)
//   (b) every thread solves rows below the block against it (column sweep, row in registers)
//   (c) rank-NB trailing update C -= P P^T with v_mfma_f64_16x16x4_f64, 32x32 per wave
// INLDS: the whole front sits in LDS (F, ld).  Otherwise F is the frontal array in HBM/L2 and the panel is
// staged in LDS (`panel`, leading dimension ldp) for (a)-(c).  The right-hand side is row R-3 of the front,
// so the forward substitution happens here as well.
// ------------------------------------------------------------------------------------------------------
typedef double d4_t __attribute__((ext_vector_type(4)));
#ifndef PANEL_FN
#define PANEL_FN __forceinline__
#endif

// (a) of factor_front_blocked: Cholesky of the w x w diagonal block of the LDS panel Pn (column-major, ldp),
// executed by ONE wave: lane r keeps row r in registers, pivots and multipliers travel through v_readlane.
// Writes L back into the panel and the padded block Ld[p*(NB+1)+c] (1/L[c][c] on the diagonal) for (b).
// Kept out of line: inlined next to (b) the register allocator spills both.
__device__ PANEL_FN void panel_diag_factor(double *__restrict__ Pn, int ldp, int w, double *__restrict__ Ld, int *bad) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 31;
    double D[NB];
#pragma unroll
    for (int c = 0; c < NB; c++) {
        const double v = Pn[(size_t)(c < w ? c : 0) * ldp + (r < w ? r : 0)];
        D[c] = (r < w && c < w && c <= r) ? v : ((r == c) ? 1.0 : 0.0);
    }
    int isbad = 0;
    double myinv = 1.0;
#pragma unroll
    for (int j = 0; j < NB; j++) {
        const double djj = readlane_d(D[j], j);
        if (!(djj > 0)) isbad = 1;
        const double inv = fast_rsqrt(djj), dj = djj * inv;
        D[j] = (r == j) ? dj : ((r > j) ? D[j] * inv : D[j]);
        if (r == j) myinv = inv;
#pragma unroll
        for (int c = j + 1; c < NB; c++) {
            const double lcj = readlane_d(D[j], c);
            if (r >= c) D[c] = fma(-D[j], lcj, D[c]);
        }
        __builtin_amdgcn_sched_barrier(0);       // bound the SGPR pressure of the readlane broadcasts
    }
    if (lane < 32) {
#pragma unroll
        for (int c = 0; c < NB; c++) Ld[r * (NB + 1) + c] = (c < r) ? D[c] : ((c == r) ? myinv : 0.0);
        if (r < w) {
#pragma unroll
            for (int c = 0; c < NB; c++) if (c <= r && c < w) Pn[(size_t)c * ldp + r] = D[c];
        }
        if (isbad && r == 0) *bad = 1;
    }
}
// (b): ONE row of the panel (relative index `row`, or nothing if row >= r1) solved against the factored block:
// y L^T = x as a column sweep with the row in registers; L arrives as wave-wide LDS broadcasts.  Deliberately
// not a loop over rows: with a loop LICM hoists all 528 block entries into registers (-> spills), and
// `volatile` reads serialise on lgkmcnt.  Loads/stores of columns >= w are redirected to column 0 with selects
// (32 conditional stores make hipcc spill as well).
__device__ PANEL_FN void panel_row_solve(double *__restrict__ Pn, int ldp, int w, int row, int r1, const double *__restrict__ Ld) {
    if (row < r1) {
        double y[NB];
        double *__restrict__ pr = Pn + row;
#pragma unroll
        for (int c = 0; c < NB; c++) { const double v = pr[(size_t)(c < w ? c : 0) * ldp]; y[c] = (c < w) ? v : 0.0; }
#pragma unroll
        for (int c = 0; c < NB; c++) {
            y[c] *= Ld[c * (NB + 1) + c];
#pragma unroll
            for (int p = 0; p < NB; p++) if (p > c) y[p] = fma(-y[c], Ld[p * (NB + 1) + c], y[p]);
        }
#pragma unroll
        for (int c = NB - 1; c >= 0; c--) pr[(size_t)(c < w ? c : 0) * ldp] = (c < w) ? y[c] : y[0];
    }
}

template <bool INLDS, int NT>
__device__ __forceinline__ void factor_front_blocked(double *__restrict__ F, int ld, int R, int C, int ns, double *__restrict__ panel,
                                                     double *__restrict__ invd, int *bad, long long *pf = nullptr) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int Rv = R - 2;                                   // valid rows (rhs row included, 2 pad rows excluded)
    for (int k0 = 0; k0 < ns; k0 += NB) {
        const int w = min(NB, ns - k0), c0 = k0 + w;
        const int ldp = INLDS ? ld : ((R - k0) | 1);
        double *__restrict__ Pn = INLDS ? F + (size_t)k0 * ld + k0 : panel;      // Pn[c * ldp + (row - k0)]
        if (!INLDS) {
            const int nr = R - k0;
            for (int e = tid; e < nr * w; e += NT) { int c = e / nr, r = e - c * nr; Pn[(size_t)c * ldp + r] = F[(size_t)(k0 + c) * ld + k0 + r]; }
            __syncthreads();
        }
        // (a) diagonal block on wave 0
        long long ts0 = 0, ts1 = 0, ts2 = 0;
        if (pf && tid == 0) ts0 = wall_clock64();
        if (wave == 0) panel_diag_factor(Pn, ldp, w, invd, bad);
        __syncthreads();
        if (pf && tid == 0) ts1 = wall_clock64();
        // (b) rows below the diagonal block: y L^T = x
        panel_row_solve(Pn, ldp, w, c0 - k0 + tid, Rv - k0, invd);
        if (Rv - c0 > NT) panel_row_solve(Pn, ldp, w, c0 - k0 + NT + tid, Rv - k0, invd);      // fronts are capped at 2*NT rows
        __syncthreads();
        if (pf && tid == 0) ts2 = wall_clock64();
        if (!INLDS) {       // the finished L panel goes back to the frontal array
            const int nr = Rv - k0;
            for (int e = tid; e < nr * w; e += NT) { int c = e / nr, r = e - c * nr; F[(size_t)(k0 + c) * ld + k0 + r] = Pn[(size_t)c * ldp + r]; }
        }
        // (c) trailing update, 32x32 tiles (ti >= tj) over rows [c0, Rv) x cols [c0, C)
        const int ntr = (Rv - c0 + 31) >> 5, ntc = (C - c0 + 31) >> 5;
        const int ntile = ntc * ntr - ntc * (ntc - 1) / 2;
        const int l15 = lane & 15, l4 = lane >> 4;
        const int nk = (w + 3) >> 2;
        for (int tl = wave; tl < ntile; tl += NT / 64) {
            int tj = 0, rem = tl;
            while (rem >= ntr - tj) { rem -= ntr - tj; tj++; }
            const int ti = tj + rem;
            const int i0 = c0 + 32 * ti, j0 = c0 + 32 * tj;
            d4_t acc[2][2];
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) acc[a][b] = (d4_t){ 0, 0, 0, 0 };
            for (int ks = 0; ks < nk; ks++) {
                const int kk = 4 * ks + l4;
                const bool kok = kk < w;
                const double *__restrict__ pk = Pn + (size_t)(kok ? kk : 0) * ldp - k0;
                double pj[2], pi[2];
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int rj = j0 + 16 * q + l15, ri = i0 + 16 * q + l15;
                    pj[q] = (kok && rj < C) ? pk[rj] : 0.0;
                    pi[q] = (kok && ri < Rv) ? pk[ri] : 0.0;
                }
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(pj[a], pi[b], acc[a][b], 0, 0, 0);
            }
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        const int i = i0 + 16 * b + l15, j = j0 + 16 * a + l4 + 4 * reg;
                        if (i < Rv && j < C && i >= j) {
                            double *p = F + (size_t)j * ld + i;
                            *p = *p - acc[a][b][reg];
                        }
                    }
        }
        __syncthreads();
        if (pf && tid == 0) { long long ts3 = wall_clock64(); pf[4] += ts1 - ts0; pf[5] += ts2 - ts1; pf[6] += ts3 - ts2; pf[7] += 1; }
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
        double i00 = fast_rsqrt(d00), l00 = d00 * i00;
        double l10 = d10 * i00, l20 = d20 * i00;
        double t11 = d11 - l10 * l10;
        double i11 = fast_rsqrt(t11), l11 = t11 * i11;
        double l21 = (d21 - l20 * l10) * i11;
        double t22 = d22 - l20 * l20 - l21 * l21;
        double i22 = fast_rsqrt(t22), l22 = t22 * i22;
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

// LDS scratch shared by both front kernels, carved from the dynamic segment after the front / panel
__device__ __forceinline__ void carve_scratch(double *base, size_t doubles_used, int R, double **invd, ChildDesc **cd, int **scratch) {
    double *p = base + doubles_used;
    *invd = p; p += NB * (NB + 1);     // factored diagonal block Ld[NB][NB+1]
    *cd = (ChildDesc *)p; p += (MAXC * sizeof(ChildDesc) + 7) / 8;
    *scratch = (int *)p;       // MAXC * R ints
}
__host__ __device__ inline size_t scratch_bytes(int R) { return NB * (NB + 1) * 8 + ((MAXC * sizeof(ChildDesc) + 7) / 8) * 8 + (size_t)MAXC * R * 4 + 16; }
__host__ __device__ inline size_t small_front_lds(int R, int C) { return (size_t)(R | 1) * C * 8 + scratch_bytes(R); }
__host__ __device__ inline size_t medium_front_lds(int R) { return (size_t)(R | 1) * NB * 8 + scratch_bytes(R); }
constexpr int TPB_MED = 512;

// small fronts: one 256-thread workgroup per front, the whole front in LDS
__global__ void __launch_bounds__(TPB) k_front_small(DevPlan P, const int *__restrict__ fronts, double *__restrict__ pool,
                                                     const double *__restrict__ Hc, int *bad) {
    extern __shared__ __attribute__((aligned(16))) double S[];
    const int t = fronts[blockIdx.x];
    const FrontDesc D = P.fd[t];
    const int nsb = D.nsb, nbc = D.nsb + D.nub;
    const int R = 3 * (nbc + 1), C = 3 * nbc, ld = R | 1;
    double *invd; ChildDesc *cd; int *scratch;
    carve_scratch(S, (size_t)ld * C, R, &invd, &cd, &scratch);
    if (P.prof && threadIdx.x == 0) P.prof[(size_t)t * 8 + 0] = wall_clock64();
    assemble_front<TPB>(P, D, pool, Hc, 0, nbc, true, S, ld, scratch, cd);
    if (P.prof && threadIdx.x == 0) P.prof[(size_t)t * 8 + 1] = wall_clock64();
    // rank-3 steps measured ~1.6x faster than the NB=32 blocked variant on LDS-resident fronts (chain-bound)
    factor_front_lds(S, ld, nsb, nbc, bad);
    if (P.prof && threadIdx.x == 0) P.prof[(size_t)t * 8 + 2] = wall_clock64();
    // store the block-lower trapezoid back to HBM (L panel + update block)
    double *Fg = pool + D.off;
    for (int c = threadIdx.x >> 6; c < C; c += TPB / 64) {
        int r0 = 3 * (c / 3);
        for (int r = r0 + (threadIdx.x & 63); r < R; r += 64) Fg[(size_t)c * R + r] = S[(size_t)c * ld + r];
    }
    if (P.prof) { __syncthreads(); if (threadIdx.x == 0) P.prof[(size_t)t * 8 + 3] = wall_clock64(); }
}

// medium fronts: one 512-thread workgroup per front; the frontal array stays in HBM/L2, panels go through LDS
__global__ void __launch_bounds__(TPB_MED) k_front_medium(DevPlan P, const int *__restrict__ fronts, double *__restrict__ pool,
                                                          const double *__restrict__ Hc, int *bad) {
    extern __shared__ __attribute__((aligned(16))) double Sm[];
    const int t = fronts[blockIdx.x];
    const FrontDesc D = P.fd[t];
    const int nbc = D.nsb + D.nub;
    const int R = 3 * (nbc + 1), C = 3 * nbc;
    double *invd; ChildDesc *cd; int *scratch;
    carve_scratch(Sm, (size_t)(R | 1) * NB, R, &invd, &cd, &scratch);
    double *Fg = pool + D.off;
    if (P.prof && threadIdx.x == 0) P.prof[(size_t)t * 8 + 0] = wall_clock64();
    assemble_front<TPB_MED>(P, D, pool, Hc, 0, nbc, true, Fg, R, scratch, cd);
    if (P.prof && threadIdx.x == 0) P.prof[(size_t)t * 8 + 1] = wall_clock64();
    factor_front_blocked<false, TPB_MED>(Fg, R, R, C, 3 * D.nsb, Sm, invd, bad, P.prof ? P.prof + (size_t)t * 8 : nullptr);
    if (P.prof && threadIdx.x == 0) { P.prof[(size_t)t * 8 + 2] = wall_clock64(); P.prof[(size_t)t * 8 + 3] = P.prof[(size_t)t * 8 + 2]; }
}

// ------------------------------------------------------------------------------------------------------
// big fronts
// ------------------------------------------------------------------------------------------------------
// work item: one chunk of ASM_CB block columns of one big front; list/pre = launch table of the level
__global__ void __launch_bounds__(TPB) k_assemble_big(DevPlan P, const int *__restrict__ list, const int *__restrict__ pre, int n,
                                                      double *__restrict__ pool, const double *__restrict__ Hc) {
    const int a = find_seg(pre, n, blockIdx.x);
    const int t = list[a], bc0 = (blockIdx.x - pre[a]) * ASM_CB;
    const FrontDesc D = P.fd[t];
    const int nbc = D.nsb + D.nub;
    const int bc1 = min(bc0 + ASM_CB, nbc);
    extern __shared__ __attribute__((aligned(16))) double Sa[];
    double *invd; ChildDesc *cd; int *scratch;
    carve_scratch(Sa, 0, 3 * (nbc + 1), &invd, &cd, &scratch);
    assemble_front<TPB>(P, D, pool, Hc, bc0, bc1, false, pool + D.off, 3 * (nbc + 1), scratch, cd);
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
    const FrontDesc D_ = P.fd[t];
    const int nsb = D_.nsb, nbc = nsb + D_.nub;
    const int R = 3 * (nbc + 1), ns = 3 * nsb;
    const int k0 = step * NB, wdt = min(NB, ns - k0);
    double *Fg = pool + D_.off;
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
        double myinv = 1.0;                 // 1 / L[r][r]
#pragma unroll
        for (int j = 0; j < NB; j++) {
            double djj = readlane_d(D[j], j);
            if (!(djj > 0)) isbad = 1;
            double inv = fast_rsqrt(djj), dj = djj * inv;
            D[j] = (r == j) ? dj : ((r > j) ? D[j] * inv : D[j]);
            if (r == j) myinv = inv;
#pragma unroll
            for (int c = j + 1; c < NB; c++) {
                double lcj = readlane_d(D[j], c);
                if (r >= c) D[c] = fma(-D[j], lcj, D[c]);
            }
        }
        if (tid < 32) {
#pragma unroll
            for (int c = 0; c < NB; c++) Ld[r][c] = (c < r) ? D[c] : ((c == r) ? myinv : 0.0);
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
__global__ void __launch_bounds__(TPB) k_syrk_big(DevPlan P, const int *__restrict__ list, const int *__restrict__ pre, int n,
                                                  int step, double *__restrict__ pool) {
    const int seg = find_seg(pre, n, blockIdx.x);
    const int t = list[seg];
    const FrontDesc D_ = P.fd[t];
    const int nsb = D_.nsb, nbc = nsb + D_.nub;
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
    double *Fg = pool + D_.off;
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
    const FrontDesc D_ = P.fd[t];
    const int nsb = D_.nsb, nub = D_.nub, nbc = nsb + nub;
    const int R = 3 * (nbc + 1), ns = 3 * nsb, m = 3 * nbc;
    double *xw = smem;                 // m doubles: x over the front's rows
    double *part = smem + m;           // NB doubles: right-hand side of the current column block
    const double *Fg = pool + D_.off;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int *rows = P.f_rows + D_.rows_begin;
    for (int e = tid; e < 3 * nub; e += TPB) xw[ns + e] = x[(size_t)3 * rows[e / 3] + e % 3];
    __syncthreads();
    for (int k1 = ns; k1 > 0; k1 -= NB) {
        const int k0 = max(0, k1 - NB), wdt = k1 - k0;
        // w_c = y_c - sum_{i >= k1} L[i,c] x[i]: each wave owns 8 of the (<= 32) columns and streams them
        // together (8 independent loads in flight per lane), lanes run down the rows (coalesced)
        {
            const int c0 = wave * 8;
            double acc[8];
            const double *colp[8];
#pragma unroll
            for (int q = 0; q < 8; q++) { acc[q] = 0; colp[q] = Fg + (size_t)(k0 + min(c0 + q, wdt - 1)) * R; }
            for (int i = k1 + lane; i < m; i += 64) {
                const double xv = xw[i];
#pragma unroll
                for (int q = 0; q < 8; q++) acc[q] = fma(colp[q][i], xv, acc[q]);
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const double sum = wave_sum(acc[q]);
                if (lane == 0 && c0 + q < wdt) part[c0 + q] = colp[q][m] - sum;     // row m = solved rhs row y
            }
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
            double dsel = 1.0;
#pragma unroll
            for (int i = 0; i < NB; i++) dsel = (c == i) ? Lc[i] : dsel;
            const double rinv = 1.0 / dsel;                  // all 32 reciprocals in parallel, off the chain
#pragma unroll
            for (int i = NB - 1; i >= 0; i--) {
                const double xi = readlane_d(w, i) * readlane_d(rinv, i);   // lane i holds the finished w_i
                if (c == i) w = xi;
                else if (c < i) w = fma(-Lc[i], xi, w);
            }
            if (lane < wdt) xw[k0 + lane] = w;
        }
        __syncthreads();
    }
    for (int e = tid; e < ns; e += TPB) x[(size_t)3 * D_.first + e] = xw[e];
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
