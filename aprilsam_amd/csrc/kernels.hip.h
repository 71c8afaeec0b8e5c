// kernels.hip.h — hand-written HIP kernels (gfx950 / CDNA4, wave64) for the Gauss-Newton hot path.
//
//   k_linearize_t      xyt / xytpos factor linearisation + 3x3 J^T W J blocks   (april_graph_xyt.c:62-124,
//                      april_graph_xytpos.c:63-102, aprilsam.c:159-192); large graphs write their results out through
//                      a wave-private LDS buffer (coalesced slot stores)                     HBM-bound streaming
//   k_front_small      one workgroup per front whose own columns fit LDS, two modes: the whole front in LDS, or only
//                      its own columns (panel mode: update columns assembled in HBM/L2, Schur update by MFMA);
//                      owner-wave extend-add of the children; elimination 16 columns at a time in registers
//                      (v_readlane pivot chain) with an MFMA update right of the block and look-ahead inside the
//                      workgroup (replaces cs_chol csparse.c:462-512); per-level launches, or ONE launch over
//                      several levels with per-front dependency flags                          latency bound
//   k_assemble_big     chunked gather-assembly of large fronts in HBM (L2 atomics as fire-and-forget adds)
//   k_block_chain      the 128 x 128 diagonal block of an outer block of a large front: four 32-column pivot chains on one wave
//                      (v_readlane), the inverses of the 32 x 32 diagonal blocks as a by-product, rows below solved and updated by six
//                      more waves on the matrix cores -- one workgroup per front                            latency bound
//   k_block_solve      the rows below that block: Y = X L11^-T panel by panel on the matrix cores, in registers
//   k_syrk_big(32)     trailing update C -= P P^T with v_mfma_f64_16x16x4_f64, once per outer block or per GROUP of outer blocks
//                      (K = 128 .. 384; 64 x 64 or 32 x 32 tiles)                                          FP64-MFMA bound
//   k_backsolve_blk    back substitution of the wide fronts, 128 columns at a time: a chain workgroup + helpers, flags
//   k_backsolve_w      x_T = L11^-T (y_T - L21^T x_struct) for fronts whose L panel fits LDS: a lane owns a column,
//                      one in-register chain per 64 columns (smatd.c:1075); multi-level or per-level launches
//   k_backsolve_gemv / k_backsolve_t   the same for all other fronts, 32 columns at a time, level by level
//   k_update_states    state = l_point + dx, theta wrap, NaN guard (april_graph_xyt.c:302-314); on the batch path this
//                      rides on the back substitution (backsolve_finish)
//   k_chi2 / k_reduce (+ k_reduce_parts)  chi^2 with the 1/2-on-xyt convention (april_graph.c:79-98), deterministic sum
//   k_scatter_host     blocks of host-evaluated (foreign-type) factors into the contribution slots
//   k_pack_update      packed Schur update of a front for the multi-GPU exchange
//   k_inc_prologue / k_inc_one / front_update_body   the incremental path: patches + linearisation of the new factors, a whole small
//                      step in one workgroup, low-rank updates of the fronts on a loop closure's root path
//   k_guard            debug option pool_guard: NaN-filled guard bands behind every frontal array
//
// The forward solve U^T y = B (smatd.c:1051) has no kernel of its own: the right-hand side rides along
// as an extra ROW of every front, so the factorisation leaves y in place.
//
// All arithmetic FP64.  Every sum has a fixed order => bit-reproducible: LDS accumulations have one owner wave per
// destination column; the L2 atomics used for HBM-resident destinations are issued by one wave per element, in program
// order, behind workgroup barriers between phases (see front_add).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
namespace asam {

constexpr int PROF_SLOTS = 16;    // debug stamps per front (DevPlan::prof)
constexpr int TPB = 256;          // threads per workgroup (4 waves)
constexpr int NB = 32;            // panel width of the big-front path
constexpr int TILE = 64;          // syrk output tile (4 waves x 32x32)
constexpr int ASM_CB = 4;         // block columns per assembly chunk (big fronts)
constexpr int OBP = 4;            // panels per outer block of the trailing update (K = OBP * NB for the wide update)

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
    int dinv0;                     // first slot (NB x NB doubles each) of this front's persistent inverse diagonal blocks, -1: none (see k_block_chain)
    int prim1, prim2;              // 1 + index (in this front's child range) of the children with the largest / second largest update block, 0: none (see k_assemble_big)
    int pad[2];
};
static_assert(sizeof(FrontDesc) == 64, "FrontDesc must stay one cache line");
// contributions of a destination: slots [src_begin, src_end) of Hc, or — for fronts regenerated by the
// incremental path — the slot ids src_idx[src_begin .. -src_end) (src_end < 0).  brow = -1: the rhs row.
struct DestRec { int brow, bcol, src_begin, src_end; };
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
    const int *src_idx;               // indirect source lists of regenerated fronts
    const double *lambda;         // per elimination position (block): Tikhonov term of its 3 diagonals
    long long *prof;              // debug: 8 wall-clock stamps (100 MHz) per front, or null
    int prof_mode;                // 1: stamps of the factorisation kernels, 2: of the back substitution
    int schur_first_nub;          // panel-mode small fronts with at least this many update blocks assemble their update columns AFTER the Schur product (0: never; see front_small_body)
    // dependency flags of ALL multi-level launches carry the STEP NUMBER (see wait_flag): epoch = the counter the first kernel of every numeric
    // phase advances (k_linearize: a batch iteration; the prologue: an incremental step).  Nothing is ever reset.  Which children a front waits for:
    //   flevel / l0 (batch sweeps): fronts below level l0 are complete before the launch starts and are not waited for;
    //   marks (incremental steps, else null): marks[t] == step number <=> front t is regenerated by THIS step's launch (written by the prologue,
    //   one kernel boundary earlier -- data, not a flag: an older value means "not in this step", its factor of an earlier step is what is wanted).
    int *epoch; const int *flevel; int l0; const int *marks;
};

// ---- work decomposition of the big-front kernels (shared by host launch tables and device decode) ----
__host__ __device__ inline int asm_chunks(int nbc) { return (nbc + ASM_CB - 1) / ASM_CB; }
// Trailing update of the big fronts, K = the panels [s_lo, s_hi) (whole outer blocks of OBP panels, clipped to the own columns):
//   wide   everything to the right of the K columns: after EVERY outer block (s_hi = s_lo + OBP), or -- fronts whose updates are large,
//          option syrk_pair_tiles -- after every SECOND one with K = both blocks (s_hi = s_lo + 2 OBP): the far part of the trailing
//          matrix is then read and written once per 256 columns instead of once per 128, and a tile's start / end latencies are
//          spread over twice the K loop;
//   ahead  (pairs only, after the first block of a pair) just the NEXT outer block's own columns, all rows below: what its diagonal
//          block and row solves need.
// Tiles of `tile` x `tile` over the lower trapezoid of columns [col_lo, col_hi), rows [col_lo, R - 2).
struct SyrkRange { int k_lo, k_hi, col_lo, col_hi, ntr, ntc; };
__host__ __device__ inline SyrkRange syrk_range(int R, int C, int ns, int s_lo, int s_hi, int tile = TILE, bool ahead = false) {
    SyrkRange g;
    g.k_lo = s_lo * NB;
    g.k_hi = s_hi * NB < ns ? s_hi * NB : ns;
    g.col_lo = g.k_hi;
    g.col_hi = ahead ? (g.k_hi + OBP * NB < ns ? g.k_hi + OBP * NB : ns) : C;
    if (g.k_hi <= g.k_lo || g.col_hi <= g.col_lo) { g.ntr = g.ntc = 0; return g; }
    g.ntr = ((R - 2) - g.col_lo + tile - 1) / tile;
    g.ntc = (g.col_hi - g.col_lo + tile - 1) / tile;
    return g;
}
__host__ __device__ inline int syrk_tiles(int R, int C, int ns, int s_lo, int s_hi, int tile = TILE, bool ahead = false) {
    SyrkRange g = syrk_range(R, C, ns, s_lo, s_hi, tile, ahead);
    return g.ntc * g.ntr - g.ntc * (g.ntc - 1) / 2;
}
// segment a with pre[a] <= bid < pre[a+1]
__device__ __forceinline__ int find_seg(const int *__restrict__ pre, int n, int bid) {
    int lo = 0, hi = n;
    while (hi - lo > 1) { int m = (lo + hi) >> 1; if (pre[m] <= bid) lo = m; else hi = m; }
    return lo;
}

__device__ __forceinline__ void trapezoid_tile(int l, int ntr, int ntc, int *ti, int *tj_);
// The same nt tiles in an order made for the eight XCDs (each with its own L2): workgroup l runs on XCD l mod 8, and workgroups
// are dispatched in order, two per compute unit -- so the 64 tiles an XCD has in hand at any moment are 64 CONSECUTIVE ones of its
// class.  Class x takes a contiguous eighth of the "strip order": strips of SYRK_SB tile columns, row by row inside a strip;
// 64 consecutive tiles are then an 8 x 8 block whose operands are 8 + 8 panel strips (1 MB at K = 128) instead of every row
// strip of the panel (the column-major order: each XCD streamed the whole panel, 4.6 MB at the root of the 1 M lattice, once per
// seven tile columns).  A bijection of [0, nt) onto the trapezoid (solver.hip.cpp: selftest).
constexpr int SYRK_SB = 8;
constexpr int SYRK_MODE_XCD_SHIFT = 8;       // k_syrk_big's mode argument: bits 0-7 the update mode, bits 8.. the tile count from which this order is used (0: never)
__host__ __device__ inline void trapezoid_tile_xcd(int l, int nt, int ntr, int ntc, int *ti, int *tj) {
    const int q = nt >> 3, rem = nt & 7, x = l & 7;
    int lp = x * q + (x < rem ? x : rem) + (l >> 3);
    for (int c0 = 0;; ) {
        const int w = ntc - c0 < SYRK_SB ? ntc - c0 : SYRK_SB;
        const int cnt = w * (ntr - c0) - w * (w - 1) / 2;
        if (lp < cnt || c0 + w >= ntc) {
            const int tri = w * (w + 1) / 2;
            if (lp < tri) { int r = 0; while ((r + 1) * (r + 2) / 2 <= lp) r++; *ti = c0 + r; *tj = c0 + lp - r * (r + 1) / 2; }
            else { const int e = lp - tri; *ti = c0 + w + e / w; *tj = c0 + e % w; }
            return;
        }
        lp -= cnt; c0 += w;
    }
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

// workgroup barrier that orders LDS traffic only: global loads and stores issued before it stay in flight across it,
// where __syncthreads() would wait for them (the software prefetch of k_backsolve_t, the write-backs of k_block_chain)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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
// flist != nullptr: factor ids come from the list (n = f_end - f_begin entries starting at f_begin)
// Factors of types this library does not know are evaluated on the HOST through their own
// eval() function pointer; their J^T W J blocks / J^T W r segments arrive here (33 doubles per factor: Haa, Hab,
// Hbb row-major, ga, gb) and are written into the same contribution slots k_linearize uses, in the same orientation.
__global__ void __launch_bounds__(TPB) k_scatter_host(int nh, const int *__restrict__ idx, const double *__restrict__ hostH, const int *__restrict__ fb,
                                                      const unsigned char *__restrict__ swp, const int *__restrict__ slot_blk,
                                                      const int *__restrict__ slot_rhs, double *__restrict__ Hc) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nh) return;
    const int f = idx[k];
    const double *H = hostH + (size_t)33 * k;
    double *o = Hc + (size_t)slot_blk[3 * f] * 9;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) o[i * 3 + j] = (i <= j) ? H[i * 3 + j] : H[j * 3 + i];
    double *go = Hc + (size_t)slot_rhs[2 * f] * 9;
    go[0] = H[27]; go[1] = H[28]; go[2] = H[29];
    if (fb[f] >= 0) {
        const bool s = swp[f] & 1;
        double *o1 = Hc + (size_t)slot_blk[3 * f + 1] * 9, *o2 = Hc + (size_t)slot_blk[3 * f + 2] * 9;
        double *g1 = Hc + (size_t)slot_rhs[2 * f + 1] * 9;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                o1[i * 3 + j] = s ? H[9 + j * 3 + i] : H[9 + i * 3 + j];
                o2[i * 3 + j] = (i <= j) ? H[18 + i * 3 + j] : H[18 + j * 3 + i];
            }
        g1[0] = H[30]; g1[1] = H[31]; g1[2] = H[32];
    }
}

// STAGED (large graphs, chosen by the host): a thread's 33 results would leave as 33 stores whose 64 lanes hit 64 different
// cache lines (slots are 72 bytes apart at best) -- the request rate of the L2, not its bandwidth, bounds the kernel.  The
// staged form passes the results through a per-wave LDS buffer in three phases ((a,a) block + rhs a, off-diagonal block,
// (b,b) block + rhs b) and writes them out with consecutive lanes on consecutive doubles of a slot: a store instruction
// covers ~5 factors x (72 + 24 contiguous bytes) instead of 64 x 8 scattered bytes.  Wave-private, no barrier.  The direct
// form stays for small graphs, where the kernel is one latency chain and the LDS round trips only add to it.
// waves per SIMD k_linearize is compiled for.  The staged form needs 134 VGPRs when left alone (the f64 sincos) -- three waves per SIMD; pinned
// to four (128 VGPRs, no spill) it is 0.67 instead of 0.79 ms on the 1 M lattice: the regression of round 5, which pushed it over 128.  Five
// (102 VGPRs): spills, 0.85 ms.
#ifndef LIN_WAVES
#define LIN_WAVES 4
#endif
constexpr int LIN_STRIDE = 13;     // doubles per lane in the staging buffer (12 used; odd: conflict-free)
// linearise_factor: the body, one thread per factor.  gtid = the thread's index in the launch (factor f_begin + gtid); stg / sid:
// the staging buffers of the STAGED form (unused otherwise).
template <bool STAGED>
__device__ __forceinline__ void linearise_factor(int gtid, int f_begin, int f_end, const int *__restrict__ flist, const int *__restrict__ fa, const int *__restrict__ fb,
                                                 const double *__restrict__ Z, const double *__restrict__ Wm,
                                                 const double *__restrict__ lp, const double *__restrict__ st,
                                                 const unsigned char *__restrict__ swp, const int *__restrict__ slot_blk,
                                                 const int *__restrict__ slot_rhs, double *__restrict__ Hc, const double *__restrict__ upt,
                                                 double (*stg)[STAGED ? 64 * LIN_STRIDE : 1], int (*sid)[STAGED ? 128 : 1], double *mirror = nullptr) {
    int f = f_begin + gtid;
    bool active = f < f_end;
    if (!STAGED && !active) return;
    if (!active) f = f_begin;
    if (flist) f = flist[f];
    int a = fa[f], b = fb[f];
    if (a < 0) { if (!STAGED) return; active = false; a = 0; b = -1; }
    double w[9], z[3], J0[9], J1[9], r[3], pa[3], pb[3] = { 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < 9; k++) w[k] = Wm[(size_t)9 * f + k];
#pragma unroll
    for (int k = 0; k < 3; k++) z[k] = Z[(size_t)3 * f + k];
    const bool binary = b >= 0;
    // unary factors: at `st`, or -- when an incremental re-plan linearises old factors again -- at the point recorded when
    // the factor entered the system (upt, 3 doubles per factor: the node's state of that call, april_graph_xytpos.c:83-85)
    const double *src = binary ? lp + (size_t)3 * a : (upt ? upt + (size_t)3 * f : st + (size_t)3 * a);
#pragma unroll
    for (int k = 0; k < 3; k++) pa[k] = src[k];
    if (binary) {
#pragma unroll
        for (int k = 0; k < 3; k++) pb[k] = lp[(size_t)3 * b + k];
    }
    factor_residual(binary, pa, pb, z, J0, J1, r);
    // one phase of the staged write-out: v[0..9) -> slot s0, v[9..NV) -> the first doubles of slot s1 (-1: nothing)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    auto flush = [&](const double *v, auto nv_, int s0, int s1) {
        constexpr int NV = decltype(nv_)::value;
        double *mine = stg[wave] + lane * LIN_STRIDE;
#pragma unroll
        for (int k = 0; k < NV; k++) mine[k] = v[k];
        sid[wave][2 * lane] = s0; sid[wave][2 * lane + 1] = s1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // wave-private buffer: the wave's own LDS operations are in order
#pragma unroll
        for (int it = 0; it < NV; it++) {
            const int idx = it * 64 + lane, t = idx / NV, e = idx - t * NV;
            const int sl = sid[wave][2 * t + (e < 9 ? 0 : 1)];
            if (sl >= 0) Hc[(size_t)sl * 9 + (e < 9 ? e : e - 9)] = stg[wave][t * LIN_STRIDE + e];
        }
        asm volatile("" ::: "memory");
    };
    double JtW0[9], H[9], g[3], out[12];
    at_b(J0, w, JtW0);                       // J0^T W            (aprilsam.c:162)
    a_b(JtW0, J0, H);                        // (J0^T W) J0       (aprilsam.c:167)
    a_v(JtW0, r, g);                         // (J0^T W) r        (aprilsam.c:184-187)
    if constexpr (STAGED) {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) out[i * 3 + j] = (i <= j) ? H[i * 3 + j] : H[j * 3 + i];
        out[9] = g[0]; out[10] = g[1]; out[11] = g[2];
        flush(out, std::integral_constant<int, 12>{}, active ? slot_blk[3 * f] : -1, active ? slot_rhs[2 * f] : -1);
    } else {
        double *o = Hc + (size_t)slot_blk[3 * f] * 9;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) o[i * 3 + j] = (i <= j) ? H[i * 3 + j] : H[j * 3 + i];
        double *go = Hc + (size_t)slot_rhs[2 * f] * 9;
        go[0] = g[0]; go[1] = g[1]; go[2] = g[2];
        if (mirror) {           // k_inc_one: the same 33 values for tail_refactor, in LDS: [H_aa 9][g_a 3][off-diagonal 9][H_bb 9][g_b 3]
#pragma unroll
            for (int k = 0; k < 9; k++) mirror[33 * gtid + k] = o[k];
#pragma unroll
            for (int k = 0; k < 3; k++) mirror[33 * gtid + 9 + k] = g[k];
        }
    }
    if (STAGED ? __ballot(binary && active) != 0ull : binary) {
        const bool on = binary && active;
        a_b(JtW0, J1, H);                    // (J0^T W) J1: rows a, cols b
        const unsigned sw = swp[f];
        const bool s = sw & 1;
        double JtW1[9], H2[9];
        at_b(J1, w, JtW1);
        a_b(JtW1, J1, H2);
        a_v(JtW1, r, g);
        if (sw & 2) {
            // W is not symmetric as given and the reference eliminates b before a: the block it accumulates is (J1^T W) J0 at (b, a)
            // (aprilsam.c:171,520 -- upper triangle of its own order only); rows a, cols b of the symmetric system = its transpose
            double Hba[9];
            a_b(JtW1, J0, Hba);
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) H[i * 3 + j] = Hba[j * 3 + i];
        }
        if constexpr (STAGED) {
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) out[i * 3 + j] = s ? H[j * 3 + i] : H[i * 3 + j];
            flush(out, std::integral_constant<int, 9>{}, on ? slot_blk[3 * f + 1] : -1, -1);
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) out[i * 3 + j] = (i <= j) ? H2[i * 3 + j] : H2[j * 3 + i];
            out[9] = g[0]; out[10] = g[1]; out[11] = g[2];
            flush(out, std::integral_constant<int, 12>{}, on ? slot_blk[3 * f + 2] : -1, on ? slot_rhs[2 * f + 1] : -1);
        } else {
            double *o1 = Hc + (size_t)slot_blk[3 * f + 1] * 9, *o2 = Hc + (size_t)slot_blk[3 * f + 2] * 9;
            double *g1 = Hc + (size_t)slot_rhs[2 * f + 1] * 9;
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) o1[i * 3 + j] = s ? H[j * 3 + i] : H[i * 3 + j];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) o2[i * 3 + j] = (i <= j) ? H2[i * 3 + j] : H2[j * 3 + i];
            g1[0] = g[0]; g1[1] = g[1]; g1[2] = g[2];
            if (mirror) {
#pragma unroll
                for (int k = 0; k < 9; k++) { mirror[33 * gtid + 12 + k] = o1[k]; mirror[33 * gtid + 21 + k] = o2[k]; }
#pragma unroll
                for (int k = 0; k < 3; k++) mirror[33 * gtid + 30 + k] = g[k];
            }
        }
    }
}

template <bool STAGED>
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(LIN_WAVES, LIN_WAVES))) k_linearize_t(int f_begin, int f_end, const int *__restrict__ flist, const int *__restrict__ fa, const int *__restrict__ fb,
                                                   const double *__restrict__ Z, const double *__restrict__ Wm,
                                                   const double *__restrict__ lp, const double *__restrict__ st,
                                                   const unsigned char *__restrict__ swp, const int *__restrict__ slot_blk,
                                                   const int *__restrict__ slot_rhs, double *__restrict__ Hc, int *__restrict__ bad = nullptr,
                                                   const double *__restrict__ upt = nullptr, int *__restrict__ epoch = nullptr) {
    __shared__ double stg[STAGED ? TPB / 64 : 1][STAGED ? 64 * LIN_STRIDE : 1];
    __shared__ int sid[STAGED ? TPB / 64 : 1][STAGED ? 128 : 1];
    if (bad && blockIdx.x == 0 && threadIdx.x == 0) { bad[0] = 0; bad[1] = 0; bad[2] = 0; bad[3] = 0; }   // "not positive definite" record of this iteration
    // dependency flags of the multi-level launches (kernels below) carry the step number and are never reset: this kernel only advances the
    // counter, one kernel boundary before the first launch that reads it
    if (epoch && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(epoch, 1);
    linearise_factor<STAGED>(blockIdx.x * blockDim.x + threadIdx.x, f_begin, f_end, flist, fa, fb, Z, Wm, lp, st, swp, slot_blk, slot_rhs, Hc, upt, stg, sid);
}

// debug / stage-level parity: the sum of every destination record exactly as assemble_front's step 2 forms it
// (same source lists, same order), 9 doubles per record
__global__ void __launch_bounds__(TPB) k_debug_dest(int nd, const DestRec *__restrict__ dest, const int *__restrict__ src_idx,
                                                    const double *__restrict__ Hc, double *__restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nd * 9) return;
    const int d = e / 9, k = e % 9;
    const DestRec rec = dest[d];
    double acc = 0;
    if (rec.src_end >= 0) { for (int q = rec.src_begin; q < rec.src_end; q++) acc += Hc[(size_t)q * 9 + k]; }
    else { for (int q = rec.src_begin; q < -rec.src_end; q++) acc += Hc[(size_t)src_idx[q] * 9 + k]; }
    out[e] = acc;
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
// ... of MANY doubles (the 10^6-pose lattice has 3 x 10^6 factors: one workgroup read them at 19 GB/s, 1.3 ms -- fifteen times k_chi2 itself):
// REDUCE_PARTS workgroups sum one contiguous chunk each into parts[], k_reduce adds the parts.  The chunks and both trees depend on n alone, so the
// sum is as reproducible as the one-stage form; sums of at most REDUCE_SPLIT terms keep that form (and their bits: M3500, the incremental demo).
constexpr int REDUCE_SPLIT = 1 << 16, REDUCE_PARTS = 1024;
__global__ void __launch_bounds__(TPB) k_reduce_parts(int n, const double *__restrict__ in, double *__restrict__ parts) {
    __shared__ double s[TPB];
    const int chunk = (n + (int)gridDim.x - 1) / (int)gridDim.x;
    const int i0 = (int)blockIdx.x * chunk, i1 = min(n, i0 + chunk);
    double acc = 0;
    for (int i = i0 + (int)threadIdx.x; i < i1; i += TPB) acc += in[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int h = TPB / 2; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) s[threadIdx.x] += s[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) parts[blockIdx.x] = s[0];
}

// ------------------------------------------------------------------------------------------------------
// gather-assembly of block columns [bc0, bc1) of front t into dst (column-major, leading dimension ld):
//   1. zero (Tikhonov lambda on own diagonals, aprilsam.c:197-204)
//   2. original J^T W J blocks and J^T W r rows, summed per destination in a fixed order
//   3. children's update matrices (extend-add), child after child (fixed order => deterministic)
// Works on LDS (small fronts: whole front or own columns) or HBM/L2 (update columns in panel mode, big fronts one chunk per
// workgroup).  NT = threads of the workgroup.  scratch: LDS ints, >= MAXC * (rows of the front) + 64.
// ------------------------------------------------------------------------------------------------------
constexpr int WCAP = 128;                // extend-add work-list entries per wave
constexpr int CAPQ = 64;                 // child records staged in LDS per front (more children: read from HBM/L2)
// LDS for the staged child records + the work lists of nw waves
__host__ __device__ constexpr int wl_bytes(int nw) { return CAPQ * 24 + nw * WCAP * 8 + nw * 64 * 8; }   // + one dummy double per lane

// accumulate into the frontal array: plain read-modify-write in LDS; in HBM/L2 a no-return FP64 atomic add executed
// by the L2 (fire and forget: no load latency on the critical path).  Every destination element is only ever
// touched by ONE wave, in program order, so the sum order -- and the bits of the result -- are fixed by the plan.
template <bool GLOBAL_DST>
__device__ __forceinline__ void front_add(double *p, double v) {
    if (GLOBAL_DST) unsafeAtomicAdd(p, v);
    else *p += v;
}

// Dependency flags of the multi-level ("persistent") launches: a front that has finished publishes its flag after a device-scope
// release fence; a dependent workgroup polls it with relaxed loads.  No deadlock by construction: the launch lists are sorted so
// that every dependency has a LOWER workgroup id, and workgroups are dispatched in id order -- the lowest unfinished workgroup
// always has all its dependencies finished, whatever else shares the GPU.  The poll is bounded all the same (a front that gives
// up flags the iteration as failed instead of hanging the device).
// WHAT a flag holds.  The STEP NUMBER -- a counter in device memory (DevPlan::epoch) that the first kernel of every numeric phase advances:
// k_linearize once per batch iteration (round 5), the prologue once per incremental step (round 6: the incremental path's launches used 0 / 1
// flags that the prologue reset every step).  Flags are never reset: a value that is read late, or from a copy of the line that is not current,
// is an OLDER step number and can only mean "not yet".  (Introduced while hunting the wrong results of profiles/r05_flag_soak.txt, whose
// cause turned out to be the release sequence -- see publish_flag; kept because it removes the reset from the protocol altogether.)  The one
// exception is k_backsolve_blk, whose chain and helper workgroups hand blocks to each other INSIDE one launch through 0 / 1 words that the chain
// resets before the launch ends (ev == 0 below): several launches of one step share the words, separated by kernel boundaries.
// Polling uses RELAXED loads (an acquire load per poll would invalidate the XCD's L2 on every iteration and slow down every
// workgroup running there); the caller then either reads the published data with agent-coherent loads (ld_agent) or
// issues ONE acquire fence for the whole workgroup (acquire_all).
// ev != 0: wait for exactly the step number; ev == 0 (k_backsolve_blk's self-resetting words): wait for any other value
__device__ __forceinline__ bool wait_flag(const int *flag, int *bad, int ev = 0) {
    int spins = 0;
    for (;;) {
        const int v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ev ? v == ev : v != 0) break;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 23)) { if (atomicCAS(bad, 0, 9) == 0) { bad[2] = 9; bad[3] = 0; } return false; }      // flag value 9: not a pivot
    }
    return true;
}
// the value a finished front publishes and its dependants wait for: the iteration number on the batch path (read once, early: the load is
// off the critical path by the time the value is needed), 0 = "flags are 0 / 1 and reset before the launch" on the other paths
__device__ __forceinline__ int flag_value(const DevPlan &P) { return P.epoch ? __hip_atomic_load(P.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0; }
// load that is coherent at agent scope without a fence: bypasses whatever stale copy of the line this XCD's L2 may hold
// (global_load ... sc1).  Data written before another workgroup's publish_flag is visible to it once the flag is seen.
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void acquire_all() {                 // after the polls: caches are shared by the workgroup, one invalidate serves all
    __syncthreads();
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
}
__device__ __forceinline__ void publish_flag(int *flag, int ev = 0) {       // call by ALL threads of the workgroup after their last store
    // every wave waits until ITS stores are in the L2: the barrier alone does not (a workgroup-scope release on gfx950 waits for no store,
    // the compute unit's vector L1 keeps its waves' stores in order among themselves) and thread 0's write-back below waits only for its own wave
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        // The release, WRITTEN OUT: write back the XCD's L2 (one write-back for the workgroup), WAIT for it, then the flag.  As an
        // __ATOMIC_RELEASE store the compiler (hipcc of ROCm 7.2, gfx950) emitted buffer_wbl2 and the store WITHOUT the s_waitcnt between them at
        // 7 of this file's 24 release sites -- wherever it had no other vector memory operation on its books at that point -- among them the
        // full-LDS branch of front_small_body: the flag could reach memory before the update block it announces, and a dependant on another XCD
        // then read what the addresses held BEFORE (the iteration or the context before).  That is the defect behind the rare wrong results
        // of chain-like graphs, whose fronts all take that branch (profiles/r05_flag_soak.txt, listings in profiles/r05_release_isa.txt;
        // tools/check_release_isa.py checks every release of the generated code).
        asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __hip_atomic_store(flag, ev ? ev : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// reset of a flag on the paths that reuse the values 0 / 1: a device-scope store like every other access to a flag word
__device__ __forceinline__ void reset_flag(int *flag, int v) { __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// MODE: ASM_LDS    - every column of the chunk goes to dstL (LDS, leading dimension ldL)
//       ASM_GLOBAL - every column goes to dstG (frontal array in HBM/L2, leading dimension ldG)
//       ASM_SPLIT  - own columns (< 3*nsb) to dstL, update columns to dstG: k_front_small's panel mode, one pass
enum { ASM_LDS = 0, ASM_GLOBAL = 1, ASM_SPLIT = 2 };
template <int NT, int MODE, bool ZERO = true>
__device__ __forceinline__ void assemble_front(const DevPlan &P, const FrontDesc &D, const double *__restrict__ pool,
                                               const double *__restrict__ Hc, int bc0, int bc1, bool full,
                                               double *__restrict__ dstL, int ldL, double *__restrict__ dstG, int ldG, int2 *__restrict__ wl,
                                               long long *pf = nullptr, const int *wait_flags = nullptr, int *bad = nullptr, int ev = 0, int skip_child = -1, int skip_child2 = -1) {
    // skip_child(2): children (index in the front's child range) whose update blocks the caller has already STORED into the destination (k_assemble_big)
    const int tid = threadIdx.x;
    const int nsb = D.nsb, nbc = D.nsb + D.nub;     // own / all block columns
    const int R = 3 * (nbc + 1);                    // rows incl. rhs block row
    const int rhs_row = 3 * nbc;
    const int csplit = MODE == ASM_LDS ? (1 << 30) : (MODE == ASM_GLOBAL ? 0 : 3 * nsb);   // columns below go to LDS
    auto put = [&](int col, int row, double v) { if (col < csplit) dstL[(size_t)col * ldL + row] = v; else dstG[(size_t)col * ldG + row] = v; };
    auto add_plain = [&](int col, int row, double v) { if (col < csplit) dstL[(size_t)col * ldL + row] += v; else dstG[(size_t)col * ldG + row] += v; };
    auto add_child = [&](int col, int row, double v) {      // col is wave-uniform
        if (col < csplit) front_add<false>(dstL + (size_t)col * ldL + row, v); else front_add<true>(dstG + (size_t)col * ldG + row, v);
    };
    constexpr int NW = NT / 64;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nch = D.ch_end - D.ch_begin;
    ChildRec *__restrict__ crs = (ChildRec *)wl;                           // CAPQ records, then the work lists
    if (tid < min(nch, CAPQ)) crs[tid] = P.child[D.ch_begin + tid];      // lands while the front is being zeroed
    // 1. zero the block-lower part of the chunk's columns: a wave per column, lanes down the rows
    //    (ZERO = false: the columns already hold the Schur product -- everything below is added to it)
    if constexpr (ZERO) {
        for (int col = 3 * bc0 + wv; col < 3 * bc1; col += NW) {
            const double lam = col < 3 * nsb ? P.lambda[D.first + col / 3] : 0.0;
            for (int row = 3 * (col / 3) + lane; row < R; row += 64) put(col, row, row == col ? lam : 0.0);
        }
    }
    __syncthreads();
    if (pf && tid == 0) pf[4] = wall_clock64();
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
            const bool rhs = rec.brow < 0;
            if (rhs ? (i > 0) : (rec.brow == rec.bcol && i < j)) continue;   // rhs: one row; diagonal block: lower part
            double acc = 0;                                                   // rhs slots keep g in [0..2]: k = j there
            if (rec.src_end >= 0) {
                // four contributions per round trip (clamped loads, predicated adds in the original order): a destination
                // rarely has more, so its sum costs one memory latency instead of one per contribution
                for (int q = rec.src_begin; q < rec.src_end; q += 4) {
                    double a[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) a[j] = Hc[(size_t)min(q + j, rec.src_end - 1) * 9 + k];
#pragma unroll
                    for (int j = 0; j < 4; j++) acc = (q + j < rec.src_end) ? acc + a[j] : acc;
                }
            }
            else { for (int q = rec.src_begin; q < -rec.src_end; q++) acc += Hc[(size_t)P.src_idx[q] * 9 + k]; }
            add_plain(3 * rec.bcol + j, rhs ? rhs_row : 3 * rec.brow + i, acc);
        }
    }
    __syncthreads();
    if (pf && tid == 0) pf[5] = wall_clock64();
    // 3. children's update matrices (extend-add).  Destination block column b belongs to wave b mod NW, which alone
    //    adds every child's contribution to it, child after child -- no two waves ever touch the same element, so
    //    there is no barrier and no atomic in LDS, and the order of the sum is fixed.
    //    (a) fill: per child, lanes test its blocks (rel = child block -> block of this front) and compact the
    //        owned ones into the wave's work list in LDS: entry = (child, child block, destination block);
    //    (b) process: two entries per trip -- 6 coalesced column loads and 2 row-map loads in flight -- lanes run
    //        down the rows of the child's columns, 64 per pass.
    int2 *__restrict__ mywl = wl + CAPQ * 3 + wv * WCAP;
    // child record as wave-uniform scalars: everything derived from it (column base pointers, row counts) then
    // lives in SGPRs and the per-lane address is a 32-bit offset from a scalar base
    // (always from LDS: a global-memory alternative behind a branch makes the compiler drain EVERY outstanding load at the
    // join, i.e. before the next item's loads are issued -- fronts with more than CAPQ children stage their records in groups)
    int g0 = 0;                                    // children [g0, g0 + CAPQ) have their records in crs
    auto rec_of = [&](int q) -> ChildRec {
        ChildRec c = crs[q - g0];
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(c.uoff & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((unsigned)(c.uoff >> 32));
        c.uoff = (long long)(((unsigned long long)hi << 32) | lo);
        c.cR = __builtin_amdgcn_readfirstlane(c.cR); c.cnu = __builtin_amdgcn_readfirstlane(c.cnu);
        c.rel_begin = __builtin_amdgcn_readfirstlane(c.rel_begin);
        return c;
    };
    constexpr int TE = 4;                                                        // items per trip: 3 * TE column loads in flight per lane (8 at 1024 threads, round 6: M3500 0.2129 -> 0.2268 ms)
    struct Trip { double v[3 * TE]; int d[TE], rc[TE], col[TE], cc[TE], nrow[TE], r0[TE]; };   // item = (child block, 64-row pass 0 or 1)
    auto load_trip = [&](int i, int n, Trip &T) {
        // three passes over the items so that the LDS reads of all of them (work-list entry, then child record) are in
        // flight together: one LDS round trip per pass instead of two per item
        int2 e[TE]; ChildRec cr[TE];
#pragma unroll
        for (int u = 0; u < TE; u++) e[u] = mywl[i + u < n ? i + u : i];
#pragma unroll
        for (int u = 0; u < TE; u++) cr[u] = crs[(__builtin_amdgcn_readfirstlane(e[u].x) & 0x3fffffff) - g0];
#pragma unroll
        for (int u = 0; u < TE; u++) {
            const bool has = i + u < n;
            const int x = __builtin_amdgcn_readfirstlane(e[u].x), y = __builtin_amdgcn_readfirstlane(e[u].y);
            ChildRec c = cr[u];                                        // work item: x = child | pass << 30, y = child block | destination block << 16
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(c.uoff & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((unsigned)(c.uoff >> 32));
            c.uoff = (long long)(((unsigned long long)hi << 32) | lo);
            c.cR = __builtin_amdgcn_readfirstlane(c.cR); c.cnu = __builtin_amdgcn_readfirstlane(c.cnu);
            c.rel_begin = __builtin_amdgcn_readfirstlane(c.rel_begin);
            const int cc = 3 * (y & 0xffff), nrow = has ? 3 * c.cnu + 1 : 0, r = cc + 64 * (x >> 30) + lane;
            T.cc[u] = cc; T.col[u] = 3 * (y >> 16); T.nrow[u] = nrow; T.r0[u] = cc + 64 * (x >> 30);
            // unconditional loads from clamped rows (no branches around memory operations: the wait counters of the
            // pipelined trips stay exact); lanes outside the child's rows / above the diagonal are masked at the add
            const double *__restrict__ sp = pool + c.uoff + (size_t)cc * c.cR;
            const int rc = has ? min(r, nrow - 1) : cc;
#pragma unroll
            for (int c3 = 0; c3 < 3; c3++) T.v[3 * u + c3] = ld_agent(sp + (size_t)c3 * c.cR + rc);
            // the row map entry stays RAW here: any arithmetic on it would make the compiler wait for this load -- and,
            // the counters being in order, for the item's column loads -- before the next item's loads are issued
            T.d[u] = P.f_rel[c.rel_begin + min(rc, nrow - 2) / 3];
            T.rc[u] = rc;
        }
    };
    // LDS adds are branch-free: lanes outside the child's rows / above the diagonal add into a private dummy slot
    double *dummy = (double *)(wl + CAPQ * 3 + NW * WCAP) + wv * 64 + lane;
    const int dummy_off = MODE == ASM_GLOBAL ? 0 : (int)(dummy - dstL);       // the dummy slot as an index from dstL (same LDS segment)
    auto apply_trip = [&](const Trip &T) {
#pragma unroll
        for (int u = 0; u < TE; u++) {
            const int r = T.r0[u] + lane;
            const bool inr = r < T.nrow[u];
            const int du = (T.rc[u] < T.nrow[u] - 1) ? 3 * T.d[u] + T.rc[u] % 3 : rhs_row;      // destination row
            if (T.col[u] < csplit) {                 // wave-uniform
#pragma unroll
                for (int c3 = 0; c3 < 3; c3++) {
                    const int o = (inr && r >= T.cc[u] + c3) ? (T.col[u] + c3) * ldL + du : dummy_off;
                    dstL[o] += T.v[3 * u + c3];
                }
            } else {
#pragma unroll
                for (int c3 = 0; c3 < 3; c3++)      // predicated: same-address atomics of masked lanes (adding 0.0) serialise in the L2 -- measured 3x slower
                    if (inr && r >= T.cc[u] + c3) unsafeAtomicAdd(dstG + (size_t)(T.col[u] + c3) * ldG + du, T.v[3 * u + c3]);
            }
        }
    };
    auto rest = [&](int i, int n) {                 // passes >= 2 of items i .. i+TE-1: children with more than 128 rows below the block
        for (int u = 0; u < TE && i + u < n; u++) {
            const int2 e = mywl[i + u];
            const int x = __builtin_amdgcn_readfirstlane(e.x), y = __builtin_amdgcn_readfirstlane(e.y);
            if (x >> 30) continue;                                    // the pass-0 item of a block carries its tail
            const ChildRec c = rec_of(x & 0x3fffffff);
            const int cc = 3 * (y & 0xffff), col = 3 * (y >> 16), nrow = __builtin_amdgcn_readfirstlane(3 * c.cnu + 1);
            if (cc + 128 >= nrow) continue;
            const double *__restrict__ sp = pool + c.uoff + (size_t)cc * c.cR;
            for (int r0 = cc + 128; r0 < nrow; r0 += 256) {         // four passes per trip: 12 loads in flight
                double v[12]; int d[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int r = r0 + 64 * k + lane;
#pragma unroll
                    for (int c3 = 0; c3 < 3; c3++) v[3 * k + c3] = ld_agent(sp + (size_t)c3 * c.cR + min(r, nrow - 1));     // clamped, masked at the add
                    d[k] = P.f_rel[c.rel_begin + min(r, nrow - 2) / 3];            // raw, see load_trip
                }
#pragma unroll
                for (int k = 0; k < 4; k++) { const int r = r0 + 64 * k + lane; d[k] = (r < nrow - 1) ? 3 * d[k] + r % 3 : rhs_row; }
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int c3 = 0; c3 < 3; c3++) if (r0 + 64 * k + lane < nrow) add_child(col + c3, d[k], v[3 * k + c3]);
            }
        }
    };
    // (two trips in flight -- the loads of trip k + 1 issued before the adds of trip k -- was measured slower: the second
    // register set costs occupancy on the wide levels and a spill in the 1024-thread variant)
    auto process = [&](int n) {
        for (int i = 0; i < n; i += TE) {
            Trip T;
            load_trip(i, n, T);
            apply_trip(T);
            rest(i, n);
        }
    };
    // (a) + (b) as ONE loop with one copy of each (code size: the instruction cache is shared by the whole CU):
    // fill until the list cannot take another 32-block slice (at most 64 items), process, resume where fill stopped.
    // The block map of the next slice is requested before the current one is compacted.
    int qn = 0, c0n = 0;                              // next slice: child qn, blocks [c0n, c0n + 32)
    auto slice_rel = [&](int q, int c0, int *cnu_out) -> int {
        if (q >= min(nch, g0 + CAPQ)) { *cnu_out = 0; return -1; }
        const ChildRec c = rec_of(q);
        if (q == skip_child || q == skip_child2) { *cnu_out = 0; return -1; }      // (a child without blocks: the fill loop moves on)
        *cnu_out = c.cnu;
        const int jb = c0 + lane;
        return (lane < 32 && jb < c.cnu) ? P.f_rel[c.rel_begin + jb] : -1;
    };
    int cnu_cur = 0;
    int rv_cur = slice_rel(0, 0, &cnu_cur);
    bool waited = false;
    for (;;) {
        const int qend = min(nch, g0 + CAPQ);
        int n = 0;
        while (qn < qend && n + 64 <= WCAP) {
            int qx = qn, cx = c0n + 32;                // the slice after this one
            if (cx >= cnu_cur) { qx++; cx = 0; }
            int cnu_nxt = 0;
            const int rv_nxt = slice_rel(qx, cx, &cnu_nxt);
            const int jb = c0n + lane, rv = rv_cur;
            const bool own = rv >= bc0 && rv < bc1 && ((rv & (NW - 1)) == wv);
            const unsigned long long mask = __ballot(own);
            if (own) mywl[n + __popcll(mask & ((1ull << lane) - 1ull))] = make_int2(qn, jb | (rv << 16));
            n += __popcll(mask);
            const bool tall = own && (3 * cnu_cur + 1 - 3 * jb > 64);       // the block has a second 64-row pass
            const unsigned long long mask2 = __ballot(tall);
            if (tall) mywl[n + __popcll(mask2 & ((1ull << lane) - 1ull))] = make_int2(qn | (1 << 30), jb | (rv << 16));
            n += __popcll(mask2);
            qn = qx; c0n = cx; rv_cur = rv_nxt; cnu_cur = cnu_nxt;
        }
        if (pf && tid == 0 && pf[6] == 0) { pf[6] = wall_clock64(); pf[7] = n; }
        // multi-level launch: everything up to here -- zero fill, factor blocks, the first work lists (static block maps) --
        // needed nothing from the children and ran while they were still being factorised; their update matrices are
        // needed from here on.  (Waiting child by child inside the extend-add, one acquire fence per wave and child, was
        // measured slower: an L2 invalidate is not cheap and hits every workgroup of the XCD.)
        if (wait_flags && !waited) {
            for (int q = tid; q < nch; q += NT) {
                const int ct = q < CAPQ ? crs[q].pad : P.child[D.ch_begin + q].pad;
                // children that are not part of this launch finished earlier: below its first level (batch sweeps), or not regenerated by this step
                if (P.marks ? __hip_atomic_load(P.marks + ct, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ev : P.flevel[ct] >= P.l0) wait_flag(wait_flags + ct, bad, ev);
            }
            __syncthreads();                       // no acquire fence: the children's update blocks are read with ld_agent
            asm volatile("" ::: "memory");
            waited = true;
            if (pf && tid == 0) pf[11] = wall_clock64();
        }
        process(n);
        if (qn >= nch) break;
        if (qn >= qend) {                           // next group of child records (every wave gets here once per group)
            __syncthreads();
            g0 += CAPQ;
            if (tid < min(nch - g0, CAPQ)) crs[tid] = P.child[D.ch_begin + g0 + tid];
            __syncthreads();
            rv_cur = slice_rel(qn, 0, &cnu_cur);
        }
    }
    __syncthreads();
}

typedef double d4_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------
// C[i][j] -= sum_{k in [k_lo, k_hi)} L[i][k] L[j][k] for the lower trapezoid j in [col_lo, col_hi), i in [j, Rv), with
// L = columns of the LDS array S (leading dimension ld).  32 x 32 tiles round-robin over the waves,
// v_mfma_f64_16x16x4_f64 fed from LDS.  C is either the same LDS array (plain read-modify-write: tiles are disjoint)
// or the frontal array in HBM/L2 (no-return L2 atomics, no load on the way).
// ------------------------------------------------------------------------------------------------------
// TO_GLOBAL: 0 = C in LDS, 1 = C in HBM/L2 by atomics, 2 = C in HBM/L2 by plain stores of the NEGATIVE product (C was nothing yet)
template <int NT, int TO_GLOBAL>
__device__ __forceinline__ void lds_panel_syrk(double *__restrict__ S, int ld, int k_lo, int k_hi, int col_lo, int col_hi, int Rv,
                                               double *__restrict__ Cg, int ldg) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    const int ntr = (Rv - col_lo + 31) / 32, ntc = (col_hi - col_lo + 31) / 32;
    const int ntiles = ntc * ntr - ntc * (ntc - 1) / 2;
    const int kw = k_hi - k_lo, nk = (kw + 3) >> 2;
    for (int l = wave; l < ntiles; l += NT / 64) {
        int ti, tj;
        trapezoid_tile(l, ntr, ntc, &ti, &tj);
        const int i0 = col_lo + 32 * ti, j0 = col_lo + 32 * tj;
        int rj[2], ri[2];
#pragma unroll
        for (int q = 0; q < 2; q++) { rj[q] = min(j0 + 16 * q + l15, Rv - 1); ri[q] = min(i0 + 16 * q + l15, Rv - 1); }
        d4_t acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) acc[a][b] = (d4_t){ 0, 0, 0, 0 };
        for (int ks = 0; ks < nk; ks++) {
            const int kk = 4 * ks + l4;
            const bool kok = kk < kw;
            const double *col = S + (size_t)(k_lo + (kok ? kk : 0)) * ld;
            double pj[2], pi[2];
#pragma unroll
            for (int q = 0; q < 2; q++) { double x = col[rj[q]], y = col[ri[q]]; pj[q] = kok ? x : 0.0; pi[q] = kok ? y : 0.0; }
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(pj[a], pi[b], acc[a][b], 0, 0, 0);
        }
        // D[a][b] element (row = l4 + 4*reg, col = l15) = update of C[i = i0+16b+l15, j = j0+16a+l4+4*reg]
        if (TO_GLOBAL) {
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        const int i = i0 + 16 * b + l15, j = j0 + 16 * a + l4 + 4 * reg;
                        if (i < Rv && j < col_hi && i >= j) {
                            if constexpr (TO_GLOBAL == 2) Cg[(size_t)j * ldg + i] = -acc[a][b][reg];
                            else unsafeAtomicAdd(Cg + (size_t)j * ldg + i, -acc[a][b][reg]);
                        }
                    }
        } else {
            // LDS: reads, then arithmetic, then writes (see factor_front_lds); out-of-range elements are redirected to a
            // valid address and written back unchanged
            double cv[16]; int off[16]; bool ok[16];
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        const int i = i0 + 16 * b + l15, j = j0 + 16 * a + l4 + 4 * reg, e = (a * 2 + b) * 4 + reg;
                        ok[e] = i < Rv && j < col_hi && i >= j;
                        off[e] = ok[e] ? j * ld + i : -1;
                    }
#pragma unroll
            for (int e = 0; e < 16; e++) cv[e] = off[e] >= 0 ? S[off[e]] : 0.0;
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) cv[(a * 2 + b) * 4 + reg] -= acc[a][b][reg];
#pragma unroll
            for (int e = 0; e < 16; e++) if (off[e] >= 0) S[off[e]] = cv[e];
        }
    }
}

// LDS -> LDS variant with 16 x 16 tiles: the fronts that live in LDS are a few tiles wide, so small tiles spread the
// update over all waves (the FP64 MFMA pipe of a SIMD, 64 cycles per instruction, is the limit here) and waste less
// above the diagonal.  K = k_hi - k_lo <= 16 (one 16-column block of the factorisation).
// Round 6: a tile of this routine was 3 000 cycles -- four dependent MFMAs and twelve LDS accesses -- because the tile index lived in a
// VGPR (the decode of the trapezoid ran per lane, in loops under exec masks), the K loop was not unrolled (two LDS round trips per
// MFMA) and the four result stores were four branches (tools/ubench/front_factor.hip: 2 900-3 450 cycles for the next block's update on
// a front of 123 rows, and far updates that outlasted the pivot chain they were meant to hide behind).  Now the tile index is a scalar
// (decoded by a few SALU instructions; one tile column when the update covers a single block), all twelve loads of a tile are issued
// before the first MFMA, the K steps are unrolled behind wave-uniform branches, and the stores are predicated.  Same operands, same
// accumulation order: bit-identical results.
template <int NT>
__device__ __forceinline__ void lds_syrk16(double *__restrict__ S, int ld, int k_lo, int k_hi, int col_lo, int col_hi, int Rv,
                                           int w0 = 0, int nw = NT / 64) {          // tiles go round-robin over waves [w0, w0 + nw)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) - w0;
    const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    if (wave < 0 || wave >= nw) return;
    const int ntr = (Rv - col_lo + 15) >> 4, ntc = (col_hi - col_lo + 15) >> 4;
    const int ntiles = ntc * ntr - ntc * (ntc - 1) / 2;
    const int kw = k_hi - k_lo, nk = (kw + 3) >> 2;                              // 1 <= nk <= 4
    for (int l = wave; l < ntiles; l += nw) {
        int ti = l, tj = 0;                                                      // column-major lower trapezoid: column tj holds ntr - tj tiles
        while (ti >= ntr - tj) { ti -= ntr - tj; tj++; }
        ti += tj;
        const int i0 = col_lo + 16 * ti, j0 = col_lo + 16 * tj;
        const int rj = min(j0 + l15, Rv - 1), ri = min(i0 + l15, Rv - 1);
        // C elements of this lane: row i = i0 + l15, columns j = j0 + l4 + 4 * reg
        const int i = i0 + l15, jb = j0 + l4;
        const int cbase = jb * ld + i;
        bool ok[4];
#pragma unroll
        for (int reg = 0; reg < 4; reg++) { const int j = jb + 4 * reg; ok[reg] = i < Rv && j < col_hi && i >= j; }
        double cv[4], x[4], y[4];
#pragma unroll
        for (int reg = 0; reg < 4; reg++) cv[reg] = S[ok[reg] ? cbase + 4 * reg * ld : 0];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const int kk = 4 * ks + l4;
            const double *col = S + (k_lo + (kk < kw ? kk : 0)) * ld;            // (clamped: always a valid address; masked below)
            x[ks] = col[rj]; y[ks] = col[ri];
        }
        d4_t acc = (d4_t){ 0, 0, 0, 0 };
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            if (ks < nk) {                                                       // (wave-uniform)
                const bool kok = 4 * ks + l4 < kw;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(kok ? x[ks] : 0.0, kok ? y[ks] : 0.0, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int reg = 0; reg < 4; reg++) if (ok[reg]) S[cbase + 4 * reg * ld] = cv[reg] - acc[reg];
    }
}

// ------------------------------------------------------------------------------------------------------
// dense right-looking Cholesky of the first nsb block columns of an LDS-resident front.
// S: column-major, leading dimension ld, nbr = nbc+1 block rows (last = rhs row + 2 zero pad rows).
// Leaves L (incl. the solved rhs row y) in the first 3*nsb columns and the Schur update in block columns < nbc_upd.
// Blocked in groups of GB block pivots (3*GB columns): inside a group the rank-3 elimination steps only touch the
// group's own columns (a few hundred 3x3 tiles: one per thread, one LDS round trip), and everything to the right
// of the group is updated ONCE per group by MFMA -- the rank-3 update of the whole trailing matrix at every pivot
// was bound by LDS instruction issue (36 ds_read/ds_write per 3x3 tile), not by the pivot chain.
// ------------------------------------------------------------------------------------------------------
constexpr int GB = 4;

// ------------------------------------------------------------------------------------------------------
// The same factorisation with the pivot chain in REGISTERS.  Columns are taken BW = 16 at a time; every wave
// holds 64 rows of the block, one row per lane, the 16 columns in registers:
//   lanes  0..15  the rows of the diagonal block itself (every wave keeps its own redundant copy),
//   lanes 16..63  48 of the rows below it (wave w: rows k0 + wdt + 48 w ...).
// The 16 scalar pivots then run without a single barrier or LDS access: pivot row entries travel by v_readlane,
// the rows below follow the same instruction stream (their triangular solve IS the same recurrence), and since
// every wave computes the diagonal block itself no wave waits for another.  Per 16 columns: one LDS load, the
// in-register chain, one LDS store, barrier, ONE MFMA update of everything to the right (K = 16), barrier —
// two barriers per 16 columns instead of two per 3 x 3 pivot (measured on M3500: 1.3 us per block pivot before).
// ------------------------------------------------------------------------------------------------------
constexpr int BW = 16;
// one block step: columns [k0, k0 + wdt), wdt <= BWT, of the LDS array S; BWT = 16, or 8 for the short last block of a front
// (a 2-column remainder should not pay for a 16-step chain)
// Dd (out): the factored diagonal block as the wave that ran chunk 0 holds it -- lane r < BWT, Dd[c] = L[k0 + r][k0 + c] for c <= r (garbage
// above the diagonal) -- for chain_store_diag AFTER the workgroup barrier that follows (other waves may still be loading their copy of
// the block from the front).  Round 6: until then the block went through a staging area in LDS and a copy by 256 threads behind the
// barrier -- 400 cycles on the critical path of every 16 columns (tools/ubench/front_factor.hip).
template <int NT, int BWT>
__device__ __forceinline__ void chain_block(double *S, int ld, int k0, int wdt, int Rv, int *bad, double (&Dd)[BWT], int nwc) {
    constexpr int BROWS = 64 - BWT;      // BROWS: rows below the diagonal block per wave; the chunks go over the waves < nwc
    constexpr int PER = BWT <= 8 ? 1 : (BWT - 2 + 7) / 8;      // independent fmas behind each of the eight steps of the next pivot's 1/sqrt (2 at BWT = 16)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool isdiag = lane < BWT;
    const int below0 = k0 + wdt;
    const int nchunk = max(1, (Rv - below0 + BROWS - 1) / BROWS);
    // (the wave's chunks from the last to the first: chunk 0 -- whose diagonal block goes out through Dd -- is then the LAST one its wave runs, and Dd
    // takes over the registers of D instead of living beside them through later chunks: with the copy made first, k_front_small<1024> spilled)
    if (wave >= nchunk) return;
    for (int ch = wave + ((nchunk - 1 - wave) / nwc) * nwc; ch >= wave; ch -= nwc) {
        const int row = isdiag ? k0 + lane : below0 + BROWS * ch + (lane - BWT);
        const bool valid = isdiag ? lane < wdt : row < Rv;
        const int rowc = valid ? row : k0;
        // plain loads, no masking: columns beyond a partial block duplicate column 0, lanes beyond the front duplicate
        // row k0, entries above the diagonal keep whatever LDS holds -- all of it stays confined to columns / lanes
        // that are never stored (updates only flow from column j to columns c > j, broadcasts only read lane c of
        // column j < c), so none of it needs to be an identity
        double D[BWT];
#pragma unroll
        for (int c = 0; c < BWT; c++) D[c] = S[(size_t)(k0 + (c < wdt ? c : 0)) * ld + rowc];
        // No predication inside the chain: entries above the diagonal (diagonal-block lanes with rr < c) turn into
        // finite garbage that nothing reads -- the broadcasts below only ever take lane c > j of column j, and the
        // store masks them.  Schedule of step j, pinned with sched_barriers (left alone the scheduler goes column by
        // column, i.e. one dependent fma chain per column): scale column j; update column j+1 FIRST and read the next
        // pivot from it; fetch all remaining pivot-row scalars (v_readlane into distinct SGPR pairs); then the
        // eight dependent operations of the next pivot's 1/sqrt, each followed by two of the independent fmas of
        // this step, so that the DP latency of the reciprocal square root hides behind the rank-1 update.
        int isbad = 0;
        double dn = readlane_d(D[0], 0);
        if (!(dn > 0)) isbad = 1;              // (wdt >= 1)
        double inv = fast_rsqrt(dn);
#pragma unroll
        for (int j = 0; j < BWT; j++) {
            D[j] *= inv;
            double lc[BWT];
            if (j + 1 < BWT) {
                lc[j + 1] = readlane_d(D[j], j + 1);
                D[j + 1] = fma(-D[j], lc[j + 1], D[j + 1]);
                dn = readlane_d(D[j + 1], j + 1);
                if (j + 1 < wdt && !(dn > 0)) isbad = 1;
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
        // rows below the block go back to LDS now; the diagonal block itself stays in the registers of the wave that ran chunk 0 and reaches
        // the front only after the barrier (other waves may still be loading their copy of it).  Stores of a partial
        // block are redirected to column 0, whose own value comes last (no conditional stores).
        if (valid && !isdiag) {
#pragma unroll
            for (int c = BWT - 1; c >= 0; c--) S[(size_t)(k0 + (c < wdt ? c : 0)) * ld + row] = (c < wdt) ? D[c] : D[0];
        }
        if (ch == 0) {
#pragma unroll
            for (int c = 0; c < BWT; c++) Dd[c] = D[c];
            if (isbad && lane == 0) { if (atomicCAS(bad, 0, 1) == 0) { bad[2] = 1; bad[3] = k0 / 3; } }
        }
    }
}
// after a workgroup barrier, by the wave that ran chunk 0 of chain_block (wave 0 of the chain's waves): the factored diagonal block goes
// from its registers into the front -- the lower triangle, and zeros above it (what the assembly left there anyway)
template <int BWT>
__device__ __forceinline__ void chain_store_diag(double *S, int ld, int k0, int wdt, const double (&Dd)[BWT]) {
    const int r = threadIdx.x & 63;
    if (r < wdt) {                                     // (one exec mask for the lot: the entries above the diagonal are stored as the zeros they are)
#pragma unroll
        for (int c = 0; c < BWT; c++) if (c < wdt) S[(k0 + c) * ld + k0 + r] = (c <= r) ? Dd[c] : 0.0;
    }
}
// Look-ahead inside the workgroup: after block k only the NEXT block's columns are updated by everybody; then the first
// few waves run block k + 1's pivot chain while the other waves apply block k to everything further right.  Every
// column still receives the updates of the blocks in order (k, then k + 1, ...), each as the same K = 16 MFMA product:
// bit-identical to the plain sequence chain / full update, two barriers per block as before, but a block now costs
// max(chain, far update) + the 16-column update instead of chain + full update.
// scalar dimensions: ns columns to eliminate, Rv valid rows, columns < hi receive the trailing update
template <int NT>
__device__ __forceinline__ void factor_dense_blk(double *S, int ld, int ns, int Rv, int hi, int *bad, double * /*unused since round 6: the staging area*/, long long *pf = nullptr) {
    constexpr int NW = NT / 64;
    long long t_chain = 0, t_syrk = 0, t0 = 0;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // one block's pivot chain on this wave's chunks of rows, the workgroup barrier behind it, and -- on wave 0, which ran chunk 0 -- the factored
    // diagonal block from its registers into the front (a handful of LDS stores that nothing on the critical path waits for).  Dd lives inside
    // this call only: declared outside, conditionally written and conditionally read, it stayed live through every other phase of the loop
    // and k_front_small<1024> spilled (1.8 MB of scratch traffic per launch in the counters).
    auto chain_sync_store = [&](int k0, int nwc) {
        const int wdt = min(BW, ns - k0);
        if (wdt <= 8) {
            double D8[8];
            chain_block<NT, 8>(S, ld, k0, wdt, Rv, bad, D8, nwc);
            __syncthreads();
            if (wave == 0) chain_store_diag<8>(S, ld, k0, wdt, D8);
        } else {
            double D16[16];
            chain_block<NT, 16>(S, ld, k0, wdt, Rv, bad, D16, nwc);
            __syncthreads();
            if (wave == 0) chain_store_diag<16>(S, ld, k0, wdt, D16);
        }
    };
    auto chain_waves = [&](int k0) {               // waves that get a chunk of rows of block k0's chain, at most half of them
        const int wdt = min(BW, ns - k0), brows = 64 - (wdt <= 8 ? 8 : 16);
        const int nchunk = max(1, (Rv - (k0 + wdt) + brows - 1) / brows);
        return min(nchunk, NW / 2);
    };
    if (pf) t0 = wall_clock64();
    chain_sync_store(0, NW);
    if (pf) { const long long t1 = wall_clock64(); t_chain += t1 - t0; t0 = t1; }
    for (int k0 = 0; k0 < ns; k0 += BW) {
        const int wdt = min(BW, ns - k0), below0 = k0 + wdt;
        if (below0 < ns) {
            const int next_hi = min(below0 + BW, ns), cw = chain_waves(below0);
            // the next block's columns, by every wave but the one that is still storing the diagonal block (when there are waves to spare)
            if (NW >= 8) lds_syrk16<NT>(S, ld, k0, below0, below0, next_hi, Rv, 1, NW - 1);
            else lds_syrk16<NT>(S, ld, k0, below0, below0, next_hi, Rv);
            __syncthreads();
            if (pf) { const long long t1 = wall_clock64(); t_syrk += t1 - t0; t0 = t1; }
            if (wave < cw) {                       // (wave-uniform: every wave meets exactly one barrier on either side)
                long long w0_ = 0, c0_ = 0;
                if (pf) { w0_ = wall_clock64(); c0_ = clock64(); }
                chain_sync_store(below0, cw);
                if (pf && threadIdx.x == 0) { pf[10] += clock64() - c0_; pf[15] += wall_clock64() - w0_; }
            } else {
                if (next_hi < hi) lds_syrk16<NT>(S, ld, k0, below0, next_hi, hi, Rv, cw, NW - cw);
                __syncthreads();
            }
            if (pf) { const long long t1 = wall_clock64(); t_chain += t1 - t0; t0 = t1; }
        } else {
            __syncthreads();
            if (below0 < hi) lds_syrk16<NT>(S, ld, k0, below0, below0, hi, Rv);
            __syncthreads();
            if (pf) { const long long t1 = wall_clock64(); t_syrk += t1 - t0; t0 = t1; }
        }
    }
    if (pf && threadIdx.x == 0) { pf[8] = t_chain; pf[9] = t_syrk; }
}
// ... in units of 3 x 3 blocks, as the fronts are described (nbc block columns + the right-hand-side row)
template <int NT>
__device__ __forceinline__ void factor_front_blk(double *S, int ld, int nsb, int nbc, int *bad, double *stage, int nbc_upd = -1, long long *pf = nullptr) {
    if (nbc_upd < 0) nbc_upd = nbc;
    factor_dense_blk<NT>(S, ld, 3 * nsb, 3 * (nbc + 1) - 2, 3 * nbc_upd, bad, stage, pf);
}

// LDS after the front / panel: the extend-add work lists
__host__ __device__ inline size_t small_front_lds(int R, int C, int nw) { return (size_t)(R | 1) * C * 8 + wl_bytes(nw); }
__host__ __device__ inline size_t panel_front_lds(int R, int ns, int nw) { return (size_t)(R | 1) * ns * 8 + wl_bytes(nw); }

// small fronts: one NT-thread workgroup per front (512 by default).  Two modes, chosen per front from its size:
//   full  — the whole frontal array in LDS (assembly, rank-3 elimination steps, one store of the block-lower part);
//   panel — only the own columns (all rows) in LDS: the update columns are assembled straight into the frontal
//           array in HBM/L2, the elimination steps touch the own columns only (left-looking with respect to the
//           update block), and the Schur update is ONE MFMA product at the end.  Covers fronts whose array
//           exceeds the LDS but whose panel fits — every front of M3500 — without the multi-launch big path.
template <int NT>
__device__ __forceinline__ void front_small_body(const DevPlan &P, const int t, double *__restrict__ pool, const double *__restrict__ Hc, int *bad,
                                                 long long full_lds_limit, int *flags, int wait, double *S) {
    const int *wflags = wait ? flags : nullptr;
    const int ev = flags ? flag_value(P) : 0;
    const FrontDesc D = P.fd[t];
    const int nsb = D.nsb, nbc = D.nsb + D.nub;
    const int R = 3 * (nbc + 1), C = 3 * nbc, ld = R | 1;
    double *Fg = pool + D.off;
    // full_lds_limit (per launch): fronts whose whole array fits run fully in LDS, the others in panel mode.  On levels with
    // far more fronts than compute units the host lowers it, trading per-front latency for workgroups per CU.
    if ((long long)small_front_lds(R, C, NT / 64) <= full_lds_limit) {
        int2 *wl = (int2 *)(S + (size_t)ld * C);
        if (P.prof && threadIdx.x == 0) P.prof[(size_t)t * PROF_SLOTS + 0] = wall_clock64();
        assemble_front<NT, ASM_LDS>(P, D, pool, Hc, 0, nbc, true, S, ld, nullptr, 0, wl, P.prof ? P.prof + (size_t)t * PROF_SLOTS : nullptr, wflags, bad, ev);
        if (P.prof && threadIdx.x == 0) P.prof[(size_t)t * PROF_SLOTS + 1] = wall_clock64();
        // rank-3 steps measured ~1.6x faster than the NB=32 blocked variant on LDS-resident fronts (chain-bound)
        factor_front_blk<NT>(S, ld, nsb, nbc, bad, (double *)wl, -1, P.prof ? P.prof + (size_t)t * PROF_SLOTS : nullptr);
        if (P.prof && threadIdx.x == 0) P.prof[(size_t)t * PROF_SLOTS + 2] = wall_clock64();
        // store the block-lower trapezoid back to HBM: the update block first -- it is all the parent waits for -- then
        // the L panel, which nobody reads before the back substitution
        for (int c = 3 * nsb + (threadIdx.x >> 6); c < C; c += NT / 64) {
            int r0 = 3 * (c / 3);
            for (int r = r0 + (threadIdx.x & 63); r < R; r += 64) Fg[(size_t)c * R + r] = S[(size_t)c * ld + r];
        }
        if (flags) publish_flag(flags + t, ev);
        for (int c = threadIdx.x >> 6; c < 3 * nsb; c += NT / 64) {
            int r0 = 3 * (c / 3);
            for (int r = r0 + (threadIdx.x & 63); r < R; r += 64) Fg[(size_t)c * R + r] = S[(size_t)c * ld + r];
        }
    } else {
        const int ns = 3 * nsb;
        int2 *wl = (int2 *)(S + (size_t)ld * ns);
        // Fronts with a large update block (round 4): the factorisation needs the OWN columns only, so the update columns are not
        // assembled first (zero fill + children's atomics, then the Schur product as atomics on top: three passes over an array
        // that lives in HBM -- 64 of the 88 us of a level-5 front of the 100 k lattice) but last: the Schur product is STORED into
        // them (no zero fill, no read-modify-write), and the factor blocks and the children's update blocks are added to it.
        const bool schur_first = P.schur_first_nub > 0 && D.nub >= P.schur_first_nub;
        if (P.prof && threadIdx.x == 0) P.prof[(size_t)t * PROF_SLOTS + 0] = wall_clock64();
        if (schur_first) assemble_front<NT, ASM_SPLIT>(P, D, pool, Hc, 0, nsb, false, S, ld, Fg, R, wl, P.prof ? P.prof + (size_t)t * PROF_SLOTS : nullptr, wflags, bad, ev);
        else assemble_front<NT, ASM_SPLIT>(P, D, pool, Hc, 0, nbc, true, S, ld, Fg, R, wl, P.prof ? P.prof + (size_t)t * PROF_SLOTS : nullptr, wflags, bad, ev);   // own columns -> LDS, update columns -> HBM/L2
        if (P.prof && threadIdx.x == 0) P.prof[(size_t)t * PROF_SLOTS + 1] = wall_clock64();
        factor_front_blk<NT>(S, ld, nsb, nbc, bad, (double *)wl, nsb, P.prof ? P.prof + (size_t)t * PROF_SLOTS : nullptr);
        if (P.prof && threadIdx.x == 0) P.prof[(size_t)t * PROF_SLOTS + 2] = wall_clock64();
        if (schur_first) {
            lds_panel_syrk<NT, 2>(S, ld, 0, ns, ns, C, R - 2, Fg, R);
            __syncthreads();       // (one compute unit, one path to the L2: the adds below arrive behind these stores)
            assemble_front<NT, ASM_GLOBAL, false>(P, D, pool, Hc, nsb, nbc, false, nullptr, 0, Fg, R, wl, nullptr, nullptr, bad);
        }
        else if (nbc > nsb) lds_panel_syrk<NT, 1>(S, ld, 0, ns, ns, C, R - 2, Fg, R);      // Schur update, K = all own columns
        if (flags) publish_flag(flags + t, ev);                                          // the parent may go; the L panel follows
        for (int c = threadIdx.x >> 6; c < ns; c += NT / 64) {
            int r0 = 3 * (c / 3);
            for (int r = r0 + (threadIdx.x & 63); r < R; r += 64) Fg[(size_t)c * R + r] = S[(size_t)c * ld + r];
        }
    }
    if (P.prof) { __syncthreads(); if (threadIdx.x == 0) P.prof[(size_t)t * PROF_SLOTS + 3] = wall_clock64(); }
}
// ------------------------------------------------------------------------------------------------------
// Low-rank UPDATE of an already factorised front (incremental path, round 4) -- the GPU counterpart of what the reference
// does row by row on the marked root paths (aprilsam.c:791-906: un-factor, add the new J^T W J, re-factor), with the cost
// matched to the change instead of to the front.  A new factor f between poses a and b adds V V^T to the normal equations and
// V c to the right-hand side, V = [J_a^T C; J_b^T C] (6 x 3), c = C^T r, W = C C^T: three vectors that enter at the front
// owning the earlier pose and travel up the assembly tree.  For a front with panel [L11; L21; y] (y: the right-hand-side row)
// and Schur update S, and incoming vectors W (one row per row of the front, the right-hand-side row included):
//     P = L11^-1 W1                         forward substitution, three right-hand sides side by side (v_readlane chain)
//     T_j = I + sum_{l<j} p_l p_l^T         3 x 3 prefix sums across lanes (wave scan), u_j = T_j^-1 p_j, g_j = sqrt(1 + p_j . u_j)
//     row i: r = W_i; for j <= i:  r -= L_ij p_j;  L'_ij = g_j L_ij + (r . u_j) / g_j      every row on its own lane, no dependency
//     W~ = (W2 - L21 P) C_T^-T,  C_T C_T^T = I + P^T P;   S' = S + W~ W~^T;   W~ goes to the parent through the child's block map
// (derivation: L'11 = L11 G with G G^T = I + P P^T taken column by column; tests/test_update_math.py checks the formulas
// against a fresh factorisation).  Several factors through one front are applied one after the other.  A front whose
// structure grew (the new pose enters the rows of every front on the path) is written to a fresh array in the new layout --
// new rows are zero rows of L -- so nothing is moved in place.  The parent only needs W~: it is published (flag) before this
// front's own update block and panel go back to HBM.
// ------------------------------------------------------------------------------------------------------
constexpr int UPD_MAXF = 4;      // new factors (3 vectors each) that may pass through the fronts of one step
constexpr int UPD_MAXC = 4;      // dirty children of one front
struct UpdRec {
    long long old_off;           // pool offset of the front's array BEFORE this step (FrontDesc::off: after; equal = same layout, in place)
    int mode;                    // 1: update (front_update_body), 0: re-assemble and re-factorise (front_small_body)
    int old_nub;                 // update blocks before this step (the own blocks do not change)
    int wout;                    // offset (doubles) of this front's outgoing vectors in the step's vector buffer: vector 3 s + q of
                                 // factor slot s at wout + (3 s + q) * (3 nub + 1), entries = update rows then the right-hand-side row
    int mask;                    // factor slots whose vectors pass through this front
    int n_own, own_f[UPD_MAXF], own_la[UPD_MAXF], own_lb[UPD_MAXF], own_slot[UPD_MAXF];      // new factors this front owns: id, local block rows of its poses (lb < 0: unary), slot
    int n_ch, ch_t[UPD_MAXC], ch_wout[UPD_MAXC], ch_rel[UPD_MAXC], ch_cnu[UPD_MAXC], ch_mask[UPD_MAXC];   // updated children: front, vectors, block map (f_rel), update blocks, slots
    int pad[3];
};
static_assert(sizeof(UpdRec) == 192, "UpdRec layout");
struct UpdCtx {                  // by-value kernel argument of the launches that may meet updated fronts (recs == nullptr: none)
    const UpdRec *recs;          // one record per entry of the launch's front list
    double *wbuf;                // the step's vector buffer
    int *wflags;                 // "outgoing vectors ready" per front (multi-level launch), or null (one workgroup, front after front)
    const int *fa, *fb; const double *Z, *Wm, *lp, *st;      // the new factors' data and linearisation points (what k_linearize reads)
};
__host__ __device__ inline size_t update_front_lds(int R, int ns, int nact) {
    return (size_t)((size_t)ns * (R | 1) + (size_t)3 * nact * (R | 1) + 8 * ns + 3 * (ns | 1) + 16) * 8;
}
template <int NT>
__device__ __forceinline__ void front_update_body(const DevPlan &P, const int t, const UpdRec &rec, const UpdCtx &uc, double *__restrict__ pool,
                                                  int *flags, int *bad, double *S) {
    constexpr int NW = NT / 64;
    const FrontDesc D = P.fd[t];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int ns = 3 * D.nsb, nu = 3 * D.nub, m = ns + nu, R = m + 3, ld = R | 1;
    const int nu_o = 3 * rec.old_nub, m_o = ns + nu_o, R_o = m_o + 3;
    const double *Fo = pool + rec.old_off;
    double *Fg = pool + D.off;
    const int nact = __popc(rec.mask), wld = nu + 1, nsp = ns | 1;
    double *Ls = S;                                  // the panel: column c at Ls[c * ld], rows 0 .. m (row m: right-hand side)
    double *Wv = Ls + (size_t)ns * ld;               // 3 nact vectors over the front's rows
    double *K8 = Wv + (size_t)3 * nact * ld;         // per own column j: p (3), u / g (3), g, -
    double *Wa = K8 + 8 * ns;                        // own rows of the current factor's vectors, consumed by the forward substitution
    double *Cm = Wa + 3 * nsp;                       // Cholesky factor of I + P^T P (6 values)
    long long *pf = P.prof ? P.prof + (size_t)t * PROF_SLOTS : nullptr;
    if (pf && tid == 0) pf[0] = wall_clock64();
    const int ev = (flags || uc.wflags) ? flag_value(P) : 0;      // the step number every flag of this launch carries
    // ---- 1. the old panel -> LDS in the new layout (zeros above the diagonal and in rows the structure gained) ----------------
    {
        const int ne = ns * R;
        for (int e0 = tid; e0 < ne; e0 += 8 * NT) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int e = min(e0 + u * NT, ne - 1), c = e / R, r = e - c * R;
                const int ro = r < m_o ? r : m_o;                    // (rows the structure gained and the pad rows: clamped, masked below)
                v[u] = Fo[(size_t)c * R_o + ro];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int e = e0 + u * NT;
                if (e < ne) {
                    const int c = e / R, r = e - c * R;
                    const bool keep = r >= c && (r < m_o || r == m);
                    Ls[(size_t)c * ld + r] = keep ? v[u] : 0.0;
                }
            }
        }
        for (int e = tid; e < 3 * nact * ld; e += NT) Wv[e] = 0.0;
    }
    __syncthreads();
    // ---- 2. incoming vectors: this front's own new factors (V = J^T C, c = C^T r) and what the updated children hand up -----------
    if (tid < rec.n_own) {
        const int f = rec.own_f[tid], la = rec.own_la[tid], lb = rec.own_lb[tid];
        const int si = __popc(rec.mask & ((1 << rec.own_slot[tid]) - 1));
        const int a = uc.fa[f], b = uc.fb[f];
        const bool binary = b >= 0;
        double w[9], z[3], J0[9], J1[9], r[3], pa[3], pb[3] = { 0, 0, 0 };
#pragma unroll
        for (int k = 0; k < 9; k++) w[k] = uc.Wm[(size_t)9 * f + k];
#pragma unroll
        for (int k = 0; k < 3; k++) z[k] = uc.Z[(size_t)3 * f + k];
        const double *src = binary ? uc.lp + (size_t)3 * a : uc.st + (size_t)3 * a;      // (april_graph_xyt.c:77-78, april_graph_xytpos.c:83-85)
#pragma unroll
        for (int k = 0; k < 3; k++) pa[k] = src[k];
        if (binary) {
#pragma unroll
            for (int k = 0; k < 3; k++) pb[k] = uc.lp[(size_t)3 * b + k];
        }
#pragma unroll
        for (int k = 0; k < 9; k++) J1[k] = 0.0;
        factor_residual(binary, pa, pb, z, J0, J1, r);
        // W = C C^T (the host only sends factors here whose W is symmetric positive definite)
        double Cc[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        Cc[0] = sqrt(w[0]); Cc[3] = w[3] / Cc[0]; Cc[6] = w[6] / Cc[0];
        Cc[4] = sqrt(w[4] - Cc[3] * Cc[3]); Cc[7] = (w[7] - Cc[6] * Cc[3]) / Cc[4];
        Cc[8] = sqrt(w[8] - Cc[6] * Cc[6] - Cc[7] * Cc[7]);
        double Va[9], Vb[9], cv[3];
        at_b(J0, Cc, Va);                            // J0^T C: rows = the pose's unknowns, columns = the three vectors
        at_b(J1, Cc, Vb);
#pragma unroll
        for (int q = 0; q < 3; q++) cv[q] = Cc[0 * 3 + q] * r[0] + Cc[1 * 3 + q] * r[1] + Cc[2 * 3 + q] * r[2];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            double *wq = Wv + (size_t)(3 * si + q) * ld;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                wq[3 * la + i] = Va[i * 3 + q];
                if (binary) wq[3 * lb + i] = Vb[i * 3 + q];
            }
            wq[m] = cv[q];
        }
    }
    // the children's block maps are static: fetched BEFORE the wait, so that one L2 round trip (the vectors) follows the flag, not two
    int pre_dst[UPD_MAXC];
#pragma unroll
    for (int k = 0; k < UPD_MAXC; k++) {
        const int kk = k < rec.n_ch ? k : 0;
        const int cnu3 = 3 * rec.ch_cnu[kk], cwld = cnu3 + 1, e = tid % cwld;
        pre_dst[k] = (k < rec.n_ch && e < cnu3) ? P.f_rel[rec.ch_rel[kk] + e / 3] : 0;
    }
    if (uc.wflags && rec.n_ch > 0) {                 // multi-level launch: the children's vectors
        if (tid < rec.n_ch) wait_flag(uc.wflags + rec.ch_t[tid], bad, ev);
        __syncthreads();
        asm volatile("" ::: "memory");
    }
    if (pf && tid == 0) pf[11] = wall_clock64();
#pragma unroll
    for (int k = 0; k < UPD_MAXC; k++) {
        if (k >= rec.n_ch) continue;
        const int cnu3 = 3 * rec.ch_cnu[k], cwld = cnu3 + 1, cm = rec.ch_mask[k];
        const double *cw = uc.wbuf + rec.ch_wout[k];
        const int *rel = P.f_rel + rec.ch_rel[k];
        for (int s = 0; s < UPD_MAXF; s++) {
            if (!((cm >> s) & 1)) continue;
            const int si = __popc(rec.mask & ((1 << s) - 1));
            for (int idx = tid; idx < 3 * cwld; idx += NT) {
                const int q = idx / cwld, e = idx - q * cwld;
                const double v = ld_agent(cw + (size_t)(3 * s + q) * cwld + e);
                const int rb = idx < NT ? pre_dst[k] : (e < cnu3 ? rel[e / 3] : 0);       // (first pass: e == tid % cwld, what the prefetch used)
                const int dst = e < cnu3 ? 3 * rb + e % 3 : m;
                Wv[(size_t)(3 * si + q) * ld + dst] = v;
            }
        }
    }
    __syncthreads();
    if (pf && tid == 0) pf[1] = wall_clock64();
    // ---- 3. factor after factor: P, the per-column scalars, then every row on its own --------------------------------------------
    for (int si = 0; si < nact; si++) {
        double *W0 = Wv + (size_t)(3 * si) * ld;
        for (int e = tid; e < 3 * ns; e += NT) { const int q = e / ns, j = e - q * ns; Wa[q * nsp + j] = W0[(size_t)q * ld + j]; }
        __syncthreads();
        // (a) P = L11^-1 W1, 32 columns at a time: a lane owns a row of the diagonal block, scaled by 1 / L[c][c]; the finished
        //     p_i travel by v_readlane; then the rows of the own part below the block take the block's products
        for (int k0 = 0; k0 < ns; k0 += 32) {
            const int wdt = min(32, ns - k0);
            if (wave == 0) {
                const int c = lane & 31;
                const bool on = c < wdt;
                const int row = k0 + (on ? c : 0);
                double l[32];
#pragma unroll
                for (int i = 0; i < 32; i++) l[i] = Ls[(size_t)(k0 + min(i, wdt - 1)) * ld + row];
                const double rinv = on ? 1.0 / Ls[(size_t)row * ld + row] : 0.0;
#pragma unroll
                for (int i = 0; i < 32; i++) l[i] = (on && i < c) ? l[i] * rinv : 0.0;       // zero at and above the diagonal: no predicate on the chain
                double s0 = on ? Wa[0 * nsp + row] * rinv : 0.0, s1 = on ? Wa[1 * nsp + row] * rinv : 0.0, s2 = on ? Wa[2 * nsp + row] * rinv : 0.0;
#pragma unroll
                for (int i = 0; i < 31; i++) {
                    const double p0 = readlane_d(s0, i), p1 = readlane_d(s1, i), p2 = readlane_d(s2, i);
                    s0 = fma(-l[i], p0, s0); s1 = fma(-l[i], p1, s1); s2 = fma(-l[i], p2, s2);
                }
                if (on && lane < 32) { K8[8 * row + 0] = s0; K8[8 * row + 1] = s1; K8[8 * row + 2] = s2; }
            }
            __syncthreads();
            const int below = k0 + wdt, nrem = ns - below;
            for (int idx = tid; idx < 3 * nrem; idx += NT) {
                const int q = idx / nrem, i = below + (idx - q * nrem);
                double acc = Wa[q * nsp + i];
                for (int j = 0; j < wdt; j++) acc = fma(-Ls[(size_t)(k0 + j) * ld + i], K8[8 * (k0 + j) + q], acc);
                Wa[q * nsp + i] = acc;
            }
            __syncthreads();
        }
        // (b) T_j = I + sum_{l<j} p_l p_l^T by a wave scan of the six distinct products, u_j = T_j^-1 p_j, g_j; I + P^T P = C C^T
        if (wave == 0) {
            double carry[6] = { 0, 0, 0, 0, 0, 0 };
            for (int j0 = 0; j0 < ns; j0 += 64) {
                const int j = j0 + lane;
                const bool on = j < ns;
                const double p0 = on ? K8[8 * j + 0] : 0.0, p1 = on ? K8[8 * j + 1] : 0.0, p2 = on ? K8[8 * j + 2] : 0.0;
                double o[6] = { p0 * p0, p0 * p1, p0 * p2, p1 * p1, p1 * p2, p2 * p2 }, own[6];
#pragma unroll
                for (int k = 0; k < 6; k++) own[k] = o[k];
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
                    for (int k = 0; k < 6; k++) { const double tt = __shfl_up(o[k], d, 64); if (lane >= d) o[k] += tt; }
                }
                const double ta = 1.0 + (o[0] - own[0] + carry[0]), tb = o[1] - own[1] + carry[1], tc = o[2] - own[2] + carry[2];
                const double td = 1.0 + (o[3] - own[3] + carry[3]), te = o[4] - own[4] + carry[4], tf = 1.0 + (o[5] - own[5] + carry[5]);
                // T = [[ta, tb, tc], [tb, td, te], [tc, te, tf]] = L D L^T
                const double l10 = tb / ta, l20 = tc / ta, d1 = td - l10 * tb, e1 = te - l20 * tb, l21 = e1 / d1, d2 = tf - l20 * tc - l21 * e1;
                const double y0 = p0, y1 = p1 - l10 * y0, y2 = p2 - l20 * y0 - l21 * y1;
                const double u2 = y2 / d2, u1 = y1 / d1 - l21 * u2, u0 = y0 / ta - l10 * u1 - l20 * u2;
                const double g = sqrt(1.0 + (p0 * u0 + p1 * u1 + p2 * u2)), gi = 1.0 / g;
                if (on) { K8[8 * j + 3] = u0 * gi; K8[8 * j + 4] = u1 * gi; K8[8 * j + 5] = u2 * gi; K8[8 * j + 6] = g; }
#pragma unroll
                for (int k = 0; k < 6; k++) carry[k] += readlane_d(o[k], 63);
            }
            if (lane == 0) {
                const double ta = 1.0 + carry[0], tb = carry[1], tc = carry[2], td = 1.0 + carry[3], te = carry[4], tf = 1.0 + carry[5];
                const double c00 = sqrt(ta), c10 = tb / c00, c20 = tc / c00, c11 = sqrt(td - c10 * c10), c21 = (te - c20 * c10) / c11;
                const double c22 = sqrt(tf - c20 * c20 - c21 * c21);
                Cm[0] = 1.0 / c00; Cm[1] = c10; Cm[2] = 1.0 / c11; Cm[3] = c20; Cm[4] = c21; Cm[5] = 1.0 / c22;
            }
        }
        __syncthreads();
        // (c) the update rows' W~ = (W2 - L21 P) C^-T first -- plain dot products, every load independent of every other -- for the
        //     block below and, after the last factor, for the parent: it needs nothing else from this front and may go
        int s = 0;
        for (int b = 0, cnt = 0; b < UPD_MAXF; b++) if ((rec.mask >> b) & 1) { if (cnt == si) s = b; cnt++; }
        {
            double *wo = uc.wbuf + rec.wout + (size_t)(3 * s) * wld;
            for (int i = ns + tid; i <= m; i += NT) {
                double r0 = W0[i], r1 = W0[(size_t)ld + i], r2 = W0[(size_t)2 * ld + i];
                const double *Li = Ls + i;
                int j = 0;
                for (; j + 4 <= ns; j += 4) {
                    double L[4], pp[12];
#pragma unroll
                    for (int u = 0; u < 4; u++) { L[u] = Li[(size_t)(j + u) * ld]; pp[3 * u] = K8[8 * (j + u)]; pp[3 * u + 1] = K8[8 * (j + u) + 1]; pp[3 * u + 2] = K8[8 * (j + u) + 2]; }
#pragma unroll
                    for (int u = 0; u < 4; u++) { r0 = fma(-L[u], pp[3 * u], r0); r1 = fma(-L[u], pp[3 * u + 1], r1); r2 = fma(-L[u], pp[3 * u + 2], r2); }
                }
                for (; j < ns; j++) { const double L = Li[(size_t)j * ld]; r0 = fma(-L, K8[8 * j], r0); r1 = fma(-L, K8[8 * j + 1], r1); r2 = fma(-L, K8[8 * j + 2], r2); }
                const double w0 = r0 * Cm[0], w1 = (r1 - Cm[1] * w0) * Cm[2], w2 = (r2 - Cm[3] * w0 - Cm[4] * w1) * Cm[5];
                wo[i - ns] = w0; wo[(size_t)wld + i - ns] = w1; wo[(size_t)2 * wld + i - ns] = w2;
            }
        }
        if (si == nact - 1 && uc.wflags) {
            if (pf && tid == 0) pf[2] = wall_clock64();
            publish_flag(uc.wflags + t, ev);         // (its barrier also separates the reads above from the writes below)
        } else __syncthreads();                      // (row r was read above by thread r - ns and is written below by thread r)
        // (d) the rows: residual and new entries, column after column (own rows stop at the diagonal); the update rows end with
        //     their W~ once more, this time kept in LDS for the update block below
        {
            for (int i = tid; i <= m; i += NT) {
                double r0 = W0[i], r1 = W0[(size_t)ld + i], r2 = W0[(size_t)2 * ld + i];
                const int jn = min(i + 1, ns);
                double *Li = Ls + i;
                int j = 0;
                for (; j + 4 <= jn; j += 4) {           // four columns per trip: their loads first (the compiler cannot move a load across the stores below)
                    double L[4], kk[28];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        L[u] = Li[(size_t)(j + u) * ld];
#pragma unroll
                        for (int v = 0; v < 7; v++) kk[7 * u + v] = K8[8 * (j + u) + v];
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        r0 = fma(-L[u], kk[7 * u], r0); r1 = fma(-L[u], kk[7 * u + 1], r1); r2 = fma(-L[u], kk[7 * u + 2], r2);
                        L[u] = fma(kk[7 * u + 6], L[u], fma(r0, kk[7 * u + 3], fma(r1, kk[7 * u + 4], r2 * kk[7 * u + 5])));
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) Li[(size_t)(j + u) * ld] = L[u];
                }
                for (; j < jn; j++) {
                    const double *k8 = K8 + 8 * j;
                    const double L = Li[(size_t)j * ld];
                    r0 = fma(-L, k8[0], r0); r1 = fma(-L, k8[1], r1); r2 = fma(-L, k8[2], r2);
                    Li[(size_t)j * ld] = fma(k8[6], L, fma(r0, k8[3], fma(r1, k8[4], r2 * k8[5])));
                }
                if (i >= ns) {
                    const double w0 = r0 * Cm[0], w1 = (r1 - Cm[1] * w0) * Cm[2], w2 = (r2 - Cm[3] * w0 - Cm[4] * w1) * Cm[5];
                    W0[i] = w0; W0[(size_t)ld + i] = w1; W0[(size_t)2 * ld + i] = w2;
                }
            }
        }
        __syncthreads();
    }
    // ---- 4. S' = S + sum W~ W~^T into the new array (lower trapezoid incl. the right-hand-side row), four columns per wave in flight ----
    for (int cu0 = 4 * wave; cu0 < nu; cu0 += 4 * NW) {
        for (int pass = 0; pass * 64 < nu + 1; pass++) {
            double old[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int cu = min(cu0 + u, nu - 1), ru = pass * 64 + lane;
                const bool has = cu < nu_o && (ru < nu_o || ru == nu);
                old[u] = has ? Fo[(size_t)(ns + cu) * R_o + ns + (ru == nu ? nu_o : ru)] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int cu = cu0 + u, ru = pass * 64 + lane;
                if (cu < nu && ru <= nu && ru >= 3 * (cu / 3)) {
                    double acc = old[u];
                    for (int v = 0; v < 3 * nact; v++) acc = fma(Wv[(size_t)v * ld + ns + ru], Wv[(size_t)v * ld + ns + cu], acc);
                    Fg[(size_t)(ns + cu) * R + ns + ru] = acc;
                }
            }
        }
    }
    // (the stores above and the ones below belong to "front done": a parent that is re-assembled reads the update block)
    for (int c = wave; c < ns; c += NW) {
        const int r0 = 3 * (c / 3);
        for (int r = r0 + lane; r < R; r += 64) Fg[(size_t)c * R + r] = Ls[(size_t)c * ld + r];
    }
    if (flags) publish_flag(flags + t, ev);
    if (pf) { __syncthreads(); if (tid == 0) pf[3] = wall_clock64(); }
}

template <int NT>
__global__ void __launch_bounds__(NT) k_front_small(DevPlan P, const int *__restrict__ fronts, double *__restrict__ pool,
                                                    const double *__restrict__ Hc, int *bad, long long full_lds_limit,
                                                    int *flags = nullptr, int wait = 0, UpdCtx uc = UpdCtx{}) {
    extern __shared__ __attribute__((aligned(16))) double S[];
    if (uc.recs) {                                   // an incremental step: some fronts of the list are updated, not re-factorised
        const UpdRec *rec = uc.recs + blockIdx.x;
        if (rec->mode) { front_update_body<NT>(P, fronts[blockIdx.x], *rec, uc, pool, flags, bad, S); return; }
    }
    front_small_body<NT>(P, fronts[blockIdx.x], pool, Hc, bad, full_lds_limit, flags, wait, S);
}

// ------------------------------------------------------------------------------------------------------
// big fronts
// ------------------------------------------------------------------------------------------------------
// work item: one chunk of ASM_CB block columns of one big front; list/pre = launch table of the level
// Round 6: STORED children.  Until now a chunk was zero-filled, then every child's update block was added to it with L2 atomics -- and on
// the lattices a level's chunks are far more than the L2 holds, so the zeros went to HBM, came back for the adds and went out again:
// 2.6-2.7 x the algorithmic bytes (profiles/r05_pmc_hbm_lattice*.json).  The two children with the largest update blocks (FrontDesc::prim1 /
// prim2, chosen at plan time; a dissection front has two children) are now STORED: one pass over the chunk writes, per element, the Tikhonov
// term or zero PLUS those children's entries where they have one (an inverse of each block map in LDS: destination block -> child block),
// coalesced down the rows, always in the order term + first + second; any further children and the factor blocks are added on top as before.
// One write per element instead of write + read + write per contributing child, and no atomics at all for a front with two children.
constexpr int ASM_INV_CAP = 4096;                  // destination block rows whose inverse maps fit the LDS (fronts with more: the zero-fill path)
__global__ void __launch_bounds__(TPB) k_assemble_big(DevPlan P, const int *__restrict__ list, const int *__restrict__ pre, int n,
                                                      double *__restrict__ pool, const double *__restrict__ Hc) {
    const int a = find_seg(pre, n, blockIdx.x);
    const int t = list[a], bc0 = (blockIdx.x - pre[a]) * ASM_CB;
    const FrontDesc D = P.fd[t];
    const int nbc = D.nsb + D.nub;
    const int bc1 = min(bc0 + ASM_CB, nbc);
    __shared__ int2 wl[wl_bytes(TPB / 64) / 8];
    __shared__ short inv[2][ASM_INV_CAP];
    const int prim = D.prim1 - 1, sec = D.prim2 - 1;
    if (prim < 0 || nbc + 1 > ASM_INV_CAP) {
        assemble_front<TPB, ASM_GLOBAL>(P, D, pool, Hc, bc0, bc1, false, nullptr, 0, pool + D.off, 3 * (nbc + 1), wl);
        return;
    }
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nst = sec >= 0 ? 2 : 1;                          // stored children
    const ChildRec cr0 = P.child[D.ch_begin + prim], cr1 = P.child[D.ch_begin + max(sec, 0)];
    const int R = 3 * (nbc + 1);
    for (int e = tid; e <= nbc; e += TPB) { inv[0][e] = -1; inv[1][e] = -1; }
    __syncthreads();
    for (int j = tid; j < cr0.cnu; j += TPB) inv[0][P.f_rel[cr0.rel_begin + j]] = (short)j;      // (a block map is strictly increasing: no two j meet)
    if (nst > 1) for (int j = tid; j < cr1.cnu; j += TPB) inv[1][P.f_rel[cr1.rel_begin + j]] = (short)j;
    if (tid == 0) { inv[0][nbc] = (short)cr0.cnu; if (nst > 1) inv[1][nbc] = (short)cr1.cnu; }     // the right-hand-side row: row 3 cnu of a child's update block
    __syncthreads();
    double *__restrict__ dst = pool + D.off;
    for (int col = 3 * bc0 + wv; col < 3 * bc1; col += TPB / 64) {
        const int bcol = col / 3, c3 = col - 3 * bcol;
        const double lam = col < 3 * D.nsb ? P.lambda[D.first + bcol] : 0.0;
        const int jc0 = __builtin_amdgcn_readfirstlane((int)inv[0][bcol]), jc1 = nst > 1 ? __builtin_amdgcn_readfirstlane((int)inv[1][bcol]) : -1;
        if (jc0 < 0 && jc1 < 0) {                              // (wave-uniform) a column neither stored child has anything for
            for (int row = 3 * bcol + lane; row < R; row += 64) dst[(size_t)col * R + row] = row == col ? lam : 0.0;
            continue;
        }
        const int cc0 = 3 * max(jc0, 0) + c3, cc1 = 3 * max(jc1, 0) + c3;
        const double *__restrict__ s0 = pool + cr0.uoff + (size_t)cc0 * cr0.cR, *__restrict__ s1 = pool + cr1.uoff + (size_t)cc1 * cr1.cR;
        for (int r0 = 3 * bcol; r0 < R; r0 += 128) {          // four loads per lane in flight
            double v0[2], v1[2]; bool ok0[2], ok1[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int row = r0 + 64 * u + lane, rc = min(row, R - 1), brow = rc / 3, i3 = rc - 3 * brow;
                // a child's entry: scalar lower triangle of its update block; its right-hand-side row (block row cnu) has one row
                const int jr0 = inv[0][brow], cr_0 = 3 * jr0 + i3, jr1 = inv[1][brow], cr_1 = 3 * jr1 + i3;
                ok0[u] = row < R && jc0 >= 0 && jr0 >= 0 && (brow < nbc ? cr_0 >= cc0 : i3 == 0);
                ok1[u] = row < R && jc1 >= 0 && jr1 >= 0 && (brow < nbc ? cr_1 >= cc1 : i3 == 0);
                v0[u] = ld_agent(s0 + (ok0[u] ? cr_0 : cc0));      // (clamped: the column's diagonal entry)
                v1[u] = ld_agent(s1 + (ok1[u] ? cr_1 : cc1));
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int row = r0 + 64 * u + lane;
                if (row < R) dst[(size_t)col * R + row] = ((row == col ? lam : 0.0) + (ok0[u] ? v0[u] : 0.0)) + (ok1[u] ? v1[u] : 0.0);
            }
        }
    }
    __syncthreads();
    assemble_front<TPB, ASM_GLOBAL, false>(P, D, pool, Hc, bc0, bc1, false, nullptr, 0, dst, R, wl, nullptr, nullptr, nullptr, 0, prim, sec);
}

// NB x NB diagonal block of a panel step, lanes = rows (lanes 32..63 mirror 0..31), columns in registers: the pinned
// schedule of chain_block (k_front_small) -- no predication inside the chain (entries above the diagonal become finite
// garbage that nothing reads: broadcasts take lane c > j of column j, the callers store c <= r only), all pivot-row
// scalars of a group of 16 columns fetched by v_readlane into distinct SGPR pairs before the fmas that use them, the
// next pivot's 1/sqrt (rsq + two Newton steps = fast_rsqrt) issued stage by stage between those fmas.  Every stored
// element sees the same operations in the same order as the predicated form.  myinv = 1 / L[r][r].
__device__ __forceinline__ void rsq_stage(int st, double dn, double &y, double &h, double &e) {   // st: compile-time constant at every call
    if (st == 0) { y = __builtin_amdgcn_rsq(dn); h = 0.5 * dn; }
    else if (st == 1 || st == 4) e = -h * y;
    else if (st == 2 || st == 5) e = fma(e, y, 0.5);
    else if (st == 3 || st == 6) y = fma(y, e, y);
}
__device__ __forceinline__ void diag_chain32(double (&D)[NB], int r, int &isbad, double &myinv) {
    double dn = readlane_d(D[0], 0);
    if (!(dn > 0)) isbad = 1;
    double inv = fast_rsqrt(dn);
#pragma unroll
    for (int j = 0; j < NB; j++) {
        D[j] *= inv;
        myinv = (r == j) ? inv : myinv;
        double y = 0, h = 0, e = 0;
        if (j + 1 < NB) {
            const double l1 = readlane_d(D[j], j + 1);
            D[j + 1] = fma(-D[j], l1, D[j + 1]);
            dn = readlane_d(D[j + 1], j + 1);
            isbad = (dn > 0) ? isbad : 1;
        }
#pragma unroll
        for (int gi = 0; gi < 2; gi++) {
            const int g = j + 2 + 16 * gi;
            if (g < NB) {
                double lc[16];
#pragma unroll
                for (int q = 0; q < 16; q++) if (g + q < NB) lc[q] = readlane_d(D[j], g + q);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < 8; p++) {
                    if (gi == 0 && p < 7) rsq_stage(p, dn, y, h, e);
#pragma unroll
                    for (int q = 2 * p; q < 2 * p + 2; q++) if (g + q < NB) D[g + q] = fma(-D[j], lc[q], D[g + q]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (j + 2 >= NB && j + 1 < NB) {
#pragma unroll
            for (int q = 0; q < 7; q++) rsq_stage(q, dn, y, h, e);
        }
        inv = y;
    }
}

// ------------------------------------------------------------------------------------------------------
// Outer-block panels (option block_panels, default): the OBP = 4 panel steps of a 128-column outer block as TWO launches
// instead of four dependent ones (102 -> 52 launches on the chain of the 100 k lattice, and far less work on it):
//   k_block_chain   one 1024-thread workgroup per front: the 128 x 128 diagonal block of the outer block goes to LDS and
//                   is factored there with the machinery of the LDS-resident fronts (16-column pivot chains in registers,
//                   MFMA updates, look-ahead inside the workgroup), goes back in place -- nobody else reads the diagonal
//                   block, so the parking + write-back of the per-panel form is gone -- and leaves the inverses of its four
//                   32 x 32 diagonal blocks for the row solves;
//   k_block_solve   the rows below it, 16 (x RB) per wave: Y = X L11^-T panel by panel, entirely on the matrix cores and in
//                   registers.  The MFMA is run "transposed" (A = rows of L / of the inverse block from LDS, B = the wave's
//                   own rows), which makes the accumulator layout of one product the B-operand layout of the next:
//                   lane (l15, l4), register r holds element (row l15, column 4 r + l4) of a 16 x 16 tile, and K-step r of a
//                   later product takes exactly that register.  Z_p = X_p - sum_{q<p} Y_q L_pq^T, Y_p = Z_p inv(L_pp)^T.
// The triangular solves against explicit 32 x 32 inverses differ from substitution by O(cond(L_pp) eps): the blocks are
// diagonal blocks of a Cholesky factor of J^T W J + lambda I, and the parity tests bound the effect (states to 1e-9).
// ------------------------------------------------------------------------------------------------------
constexpr int OBW = OBP * NB;              // columns of an outer block
constexpr int BS_LDL = 100, BS_LDI = 36;   // LDS row strides in k_block_solve (8 banks apart per row: conflict-free A-operand reads)
constexpr int BLOCK_ROWS = 64;             // rows per workgroup of k_block_solve at RB = 1 (16 per wave)
__host__ __device__ constexpr size_t block_solve_lds() { return (size_t)((OBW - NB) * BS_LDL + OBP * NB * BS_LDI) * 8; }
__host__ __device__ constexpr size_t block_chain_lds() { return block_solve_lds() + (size_t)NB * (NB + 1) * 8; }
__host__ __device__ inline int block_tiles(int R, int ns, int ob, int rb) {          // row tiles below the diagonal block of outer block ob
    const int c1 = (ob + 1) * OBW < ns ? (ob + 1) * OBW : ns;
    const int below = (R - 2) - c1;
    return below <= 0 ? 0 : (below + BLOCK_ROWS * rb - 1) / (BLOCK_ROWS * rb);
}
// k_block_chain: one workgroup per front, 7 waves at work.  Wave 0 owns the pivot chains: per 32-column panel the diagonal
// block runs through diag_chain32 (lanes 0-31 = its rows, 32 columns in registers) -- and lanes 32-63 carry the rows of the
// IDENTITY through the same instruction stream: a row below the block ends up as x L^-T, so e_i turns into row i of L^-T,
// i.e. the inverse of the block comes out of the chain for free.  Waves 1-6 own the six 16-row blocks below the first panel
// (rows 32 .. 127 of the 128 x 128 block) in MFMA layout, exactly as k_block_solve holds its rows: after panel p they solve
// their rows against it (Y_p = Z_p inv(L_pp)^T, 12 MFMAs), park them in LDS as the A operands of everybody's later updates,
// and update their later panels right-looking (Z_p' -= Y_p L_p'p^T, 16 MFMAs per panel) -- the next panel's diagonal block
// first, which goes to the chain wave through LDS while the rest of the updates runs beside the next chain.
// Critical path per panel: chain (3.3 us) + 12 + 16 MFMAs + three barriers.
constexpr int BCH_THREADS = 512;
__global__ void __launch_bounds__(BCH_THREADS) k_block_chain(DevPlan P, const int *__restrict__ list, int ob, double *__restrict__ pool,
                                                             double *__restrict__ dinv_tmp, double *__restrict__ dinv_keep, int *bad) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *Lo = sm;                                   // rows 32 .. 127 of L11, columns 0 .. 95 (k_block_solve's layout)
    double *Is = sm + (OBW - NB) * BS_LDL;             // inverse blocks: Is[(32 p + j) * BS_LDI + i] = inv(L_pp)[j][i]
    double *Dl = Is + OBP * NB * BS_LDI;               // the next panel's diagonal block on its way to the chain wave, Dl[r * (NB + 1) + c]
    const int t = list[blockIdx.x];
    const FrontDesc D_ = P.fd[t];
    const int nbc = D_.nsb + D_.nub, R = 3 * (nbc + 1), ns = 3 * D_.nsb;
    const int c0 = ob * OBW, W = min(OBW, ns - c0), np = (W + NB - 1) / NB;
    double *Fg = pool + D_.off + (size_t)c0 * R + c0;          // the diagonal block's origin
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    const int rb = wave + 1, prb = rb >> 1;            // row waves 1 .. 6: row block rb (rows 16 rb ..), whose own panel is prb
    const bool roww = wave >= 1 && wave <= 6 && 16 * rb < W;
    long long *pf = (P.prof && P.prof_mode == 3) ? P.prof + (size_t)t * PROF_SLOTS : nullptr;      // debug stamps (tools/chain_times.py)
    if (pf && tid == 0) pf[0] = wall_clock64();
    // rows of L11 that no wave owns (a partial last outer block) must read as zeros, not as whatever LDS held: they are A
    // operands of the updates, and 0 x NaN would poison the valid columns
    for (int e = tid; e < (OBW - NB) * BS_LDL; e += BCH_THREADS) Lo[e] = 0.0;
    d4_t Z[OBP][2], Y[OBP - 1][2];
    if (roww) {
        const int row = min(16 * rb + l15, W - 1);
#pragma unroll
        for (int q = 0; q < OBP; q++)
#pragma unroll
            for (int nb = 0; nb < 2; nb++)
#pragma unroll
                for (int r = 0; r < 4; r++) Z[q][nb][r] = Fg[(size_t)min(NB * min(q, prb) + 16 * nb + 4 * r + l4, W - 1) * R + row];     // (panels past the own one: never used)
    }
#pragma unroll
    for (int p = 0; p < OBP; p++) {
        if (p >= np) break;
        const int k0 = NB * p, wp = min(NB, W - k0);
        if (wave == 0) {
            const int r = lane & 31;
            double D[NB];
            if (lane < 32) {
#pragma unroll
                for (int c = 0; c < NB; c++) {
                    // (p == 0: clamped unconditional loads, masked afterwards -- a load behind a branch would be waited for on the spot)
                    const double v = (p == 0) ? Fg[(size_t)min(c, wp - 1) * R + min(r, wp - 1)] : Dl[r * (NB + 1) + c];
                    D[c] = (p > 0 || (r < wp && c <= r)) ? v : ((r == c) ? 1.0 : 0.0);
                }
            } else {
#pragma unroll
                for (int c = 0; c < NB; c++) D[c] = (r == c) ? 1.0 : 0.0;
            }
            int isbad = 0;
            double myinv = 1.0;
            if (pf && lane == 0 && p < 3) pf[1 + 5 * p] = wall_clock64();
            diag_chain32(D, r, isbad, myinv);
            if (pf && lane == 0 && p < 3) pf[2 + 5 * p] = wall_clock64();
            if (lane >= 32) {                  // lane 32 + i holds row i of L^-T: D[j] = inv(L_pp)[j][i]
#pragma unroll
                for (int j = 0; j < NB; j++) Is[(k0 + j) * BS_LDI + r] = D[j];
            }
            lds_barrier();
            // (behind the barrier: nothing on the critical path waits for these)
            if (lane < 32) {
#pragma unroll
                for (int c = 0; c < NB; c++) if (c <= r && r < wp) Fg[(size_t)(k0 + c) * R + k0 + r] = D[c];
                if (isbad && r == 0) { if (atomicCAS(bad, 0, 1) == 0) { bad[1] = t; bad[2] = 2; bad[3] = ob * OBP + p; } }
            } else {
                // kept for the back substitution when the front has a slot range of its own (k_backsolve_blk), else parked per launch slot
                double *o = D_.dinv0 >= 0 ? dinv_keep + ((size_t)D_.dinv0 + ob * OBP + p) * (NB * NB) : dinv_tmp + ((size_t)blockIdx.x * OBP + p) * (NB * NB);
#pragma unroll
                for (int j = 0; j < NB; j++) o[j * NB + r] = D[j];
            }
        } else {
            lds_barrier();
        }
        if (pf && tid == 0 && p < 3) pf[3 + 5 * p] = wall_clock64();
        if (roww && prb > p) {
            // Y_p = Z_p inv(L_pp)^T; the rows become L[16 rb .. ][k0 ..]: into the front, and into LDS for the updates
            // (three accumulators of depth 4 instead of one of depth 8: the products are latency bound)
            d4_t y[2], y1b = (d4_t){ 0, 0, 0, 0 };
            y[0] = (d4_t){ 0, 0, 0, 0 }; y[1] = (d4_t){ 0, 0, 0, 0 };
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
                const double a0 = Is[(k0 + l15) * BS_LDI + 4 * s4 + l4], a1 = Is[(k0 + 16 + l15) * BS_LDI + 4 * s4 + l4];
                const double a2 = Is[(k0 + 16 + l15) * BS_LDI + 16 + 4 * s4 + l4];
                y[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, Z[p][0][s4], y[0], 0, 0, 0);
                y[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, Z[p][0][s4], y[1], 0, 0, 0);
                y1b = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, Z[p][1][s4], y1b, 0, 0, 0);
            }
            y[1] += y1b;
            const int row = 16 * rb + l15;
#pragma unroll
            for (int jb = 0; jb < 2; jb++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int col = 16 * jb + 4 * r + l4;
                    const bool ok = row < W && col < wp;
                    Lo[(row - NB) * BS_LDL + k0 + col] = ok ? y[jb][r] : 0.0;
                    if (ok) Fg[(size_t)(k0 + col) * R + row] = y[jb][r];
                }
            if (p < OBP - 1) { Y[p][0] = y[0]; Y[p][1] = y[1]; }
        }
        lds_barrier();
        if (pf && tid == 0 && p < 3) pf[4 + 5 * p] = wall_clock64();
        // right-looking updates with panel p: Z_q -= Y_p L_qp^T for the wave's later panels q, the next panel first
        auto update = [&](auto q_) {
            constexpr int q = decltype(q_)::value;
            d4_t e[2] = { (d4_t){ 0, 0, 0, 0 }, (d4_t){ 0, 0, 0, 0 } };          // odd K-steps: four chains of depth 4
#pragma unroll
            for (int s8 = 0; s8 < 8; s8++)
#pragma unroll
                for (int nb = 0; nb < 2; nb++) {
                    const double a = -Lo[(NB * (q - 1) + 16 * nb + l15) * BS_LDL + k0 + 4 * s8 + l4];
                    const double yv = Y[p < OBP - 1 ? p : 0][s8 >> 2][s8 & 3];
                    if (s8 & 1) e[nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, yv, e[nb], 0, 0, 0);
                    else Z[q][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, yv, Z[q][nb], 0, 0, 0);
                }
            Z[q][0] += e[0]; Z[q][1] += e[1];
        };
        if (p + 1 < OBP) {
            if (roww && prb > p) {
                if (p == 0) update(std::integral_constant<int, 1>{});
                else if (p == 1) update(std::integral_constant<int, 2>{});
                else update(std::integral_constant<int, 3>{});
                if (prb == p + 1) {          // this row block belongs to the next diagonal block: lower part to the chain wave, identity beyond the front
                    const int wq = min(NB, W - NB * (p + 1));
                    const int rr = 16 * (rb & 1) + l15;
#pragma unroll
                    for (int nb = 0; nb < 2; nb++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int cc = 16 * nb + 4 * r + l4;
                            Dl[rr * (NB + 1) + cc] = (rr < wq && cc < wq && cc <= rr) ? Z[p + 1][nb][r] : ((rr == cc) ? 1.0 : 0.0);
                        }
                }
            }
            // a row block of a partial last panel that no wave owns: its part of the next diagonal block is the identity
            if (wave == 7 && p + 1 < np) {
                const int wq = min(NB, W - NB * (p + 1));
                for (int e = lane; e < NB * NB; e += 64) { const int rr = e >> 5, cc = e & 31; if (rr >= wq && (16 * (2 * (p + 1) + (rr >> 4)) >= W)) Dl[rr * (NB + 1) + cc] = (rr == cc) ? 1.0 : 0.0; }
            }
        }
        lds_barrier();
        if (pf && tid == 0 && p < 3) pf[5 + 5 * p] = wall_clock64();
        if (roww) {
            if (p == 0) { if (prb > 1) update(std::integral_constant<int, 2>{}); if (prb > 2) update(std::integral_constant<int, 3>{}); }
            else if (p == 1) { if (prb > 2) update(std::integral_constant<int, 3>{}); }
        }
    }
}

template <int RB>
__global__ void __launch_bounds__(TPB) k_block_solve(DevPlan P, const int *__restrict__ list, const int *__restrict__ pre, int n, int ob,
                                                     double *__restrict__ pool, const double *__restrict__ dinv_tmp, const double *__restrict__ dinv_keep) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *Lo = sm;                                   // rows 32 .. 127 of L11, columns 0 .. 95: Lo[(row - 32) * BS_LDL + col], zero outside the block-lower part
    double *Is = sm + (OBW - NB) * BS_LDL;             // the four inverse blocks: Is[(32 p + i) * BS_LDI + j]
    const int seg = find_seg(pre, n, blockIdx.x);
    const int t = list[seg], tile = blockIdx.x - pre[seg];
    const FrontDesc D_ = P.fd[t];
    const int nbc = D_.nsb + D_.nub, R = 3 * (nbc + 1), ns = 3 * D_.nsb, Rv = R - 2;
    const int c0 = ob * OBW, W = min(OBW, ns - c0), c1 = c0 + W, np = (W + NB - 1) / NB;
    double *Fg = pool + D_.off + (size_t)c0 * R;       // column c0 of the front
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    const int row0 = c1 + tile * (BLOCK_ROWS * RB) + wave * 16 * RB;
    // the wave's rows of all panels: requested before anything else (nothing here depends on LDS)
    d4_t X[RB][OBP][2];
#pragma unroll
    for (int rb = 0; rb < RB; rb++) {
        const int row = min(row0 + 16 * rb + l15, Rv - 1);
#pragma unroll
        for (int p = 0; p < OBP; p++)
#pragma unroll
            for (int nb = 0; nb < 2; nb++)
#pragma unroll
                for (int r = 0; r < 4; r++) X[rb][p][nb][r] = Fg[(size_t)min(NB * p + 16 * nb + 4 * r + l4, W - 1) * R + row];
    }
    {
        // the six off-diagonal 32 x 32 blocks of L11 and the four inverse blocks: all loads of a thread issued back to back
        // (24 + 16), then parked in LDS.  Rows past a partial last block are clamped duplicates: finite, and they only ever
        // reach columns that are not stored.
        double v[24], w[16];
#pragma unroll
        for (int k = 0; k < 24; k++) {
            const int e = tid + TPB * k, b = e >> 10, rr = e & 31, cc = (e >> 5) & 31;
            const int pr = b == 0 ? 1 : (b < 3 ? 2 : 3), pc = b == 0 ? 0 : (b < 3 ? b - 1 : b - 3);
            v[k] = Fg[(size_t)min(NB * pc + cc, W - 1) * R + c0 + min(NB * pr + rr, W - 1)];      // (columns past a partial last block: clamped too -- at the root front they lie past the end of the front, and of the pool)
        }
        const double *dinv = D_.dinv0 >= 0 ? dinv_keep + ((size_t)D_.dinv0 + ob * OBP) * (NB * NB) : dinv_tmp + (size_t)seg * (OBP * NB * NB);
#pragma unroll
        for (int k = 0; k < 16; k++) w[k] = dinv[tid + TPB * k];
#pragma unroll
        for (int k = 0; k < 24; k++) {
            const int e = tid + TPB * k, b = e >> 10, rr = e & 31, cc = (e >> 5) & 31;
            const int pr = b == 0 ? 1 : (b < 3 ? 2 : 3), pc = b == 0 ? 0 : (b < 3 ? b - 1 : b - 3);
            Lo[(NB * (pr - 1) + rr) * BS_LDL + NB * pc + cc] = v[k];
        }
#pragma unroll
        for (int k = 0; k < 16; k++) { const int e = tid + TPB * k; Is[(e >> 5) * BS_LDI + (e & 31)] = w[k]; }
    }
    __syncthreads();
#pragma unroll
    for (int rb = 0; rb < RB; rb++) {
        d4_t Y[OBP][2];
#pragma unroll
        for (int p = 0; p < OBP; p++) {
            if (p < np) {
                d4_t z[2] = { X[rb][p][0], X[rb][p][1] };
#pragma unroll
                for (int q = 0; q < p; q++)
#pragma unroll
                    for (int s8 = 0; s8 < 8; s8++)
#pragma unroll
                        for (int nb = 0; nb < 2; nb++) {
                            const double a = -Lo[(NB * (p - 1) + 16 * nb + l15) * BS_LDL + NB * q + 4 * s8 + l4];
                            z[nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Y[q][s8 >> 2][s8 & 3], z[nb], 0, 0, 0);
                        }
                d4_t y[2] = { (d4_t){ 0, 0, 0, 0 }, (d4_t){ 0, 0, 0, 0 } };
#pragma unroll
                for (int jb = 0; jb < 2; jb++)
#pragma unroll
                    for (int nb = 0; nb <= jb; nb++)
#pragma unroll
                        for (int s4 = 0; s4 < 4; s4++) {
                            const double a = Is[(NB * p + 16 * jb + l15) * BS_LDI + 16 * nb + 4 * s4 + l4];
                            y[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, z[nb][s4], y[jb], 0, 0, 0);
                        }
                Y[p][0] = y[0]; Y[p][1] = y[1];
            }
        }
        const int row = row0 + 16 * rb + l15;
#pragma unroll
        for (int p = 0; p < OBP; p++)
#pragma unroll
            for (int jb = 0; jb < 2; jb++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int col = NB * p + 16 * jb + 4 * r + l4;
                    if (p < np && col < W && row < Rv) Fg[(size_t)col * R + row] = Y[p][jb][r];
                }
    }
}

// trailing update of panel step `step`: C[i,j] -= sum_p P[i,p] P[j,p] for j >= k0+wdt, i >= j.
// One workgroup per 64x64 tile (w_ti >= w_tj, in units of TILE from c0 = k0 + wdt); 4 waves, each a 32x32
// quadrant as 2x2 v_mfma_f64_16x16x4_f64 tiles.  The MFMA computes the TRANSPOSED update (A = P_j rows,
// B = P_i rows) so that a lane's 4 results sit in consecutive... columns of one row run: stores coalesce
// along rows i (lanes 0-15 = 16 consecutive rows of one column).
// Q: MFMA blocks per side of a wave's quadrant -- 2: 64 x 64 tile per workgroup (TILE), 1: 32 x 32 (TILE / 2; launches with few tiles)
template <int KCH, bool CV_EARLY, int Q = 2>
__device__ __forceinline__ void syrk_big_body(const DevPlan &P, const int *__restrict__ list, const int *__restrict__ pre, int n,
                                              int s_lo, int s_hi, int mode, double *__restrict__ pool) {
    const int seg = find_seg(pre, n, blockIdx.x);
    const int t = list[seg];
    const FrontDesc D_ = P.fd[t];
    const int nsb = D_.nsb, nbc = nsb + D_.nub;
    const int R = 3 * (nbc + 1), C = 3 * nbc, ns = 3 * nsb;
    constexpr int TW = 32 * Q, QW = 16 * Q;            // tile and quadrant width
    const SyrkRange g = syrk_range(R, C, ns, s_lo, s_hi, TW, (mode & 0xff) == 2);
    const int k0 = g.k_lo, wdt = g.k_hi - g.k_lo, c0 = g.col_lo, ntr = g.ntr, ntc = g.ntc, CH = g.col_hi;
    int ti, tj;
    const int nt = ntc * ntr - ntc * (ntc - 1) / 2;
    const int xcd_min = mode >> SYRK_MODE_XCD_SHIFT;
    if (xcd_min > 0 && nt >= xcd_min) trapezoid_tile_xcd(blockIdx.x - pre[seg], nt, ntr, ntc, &ti, &tj);
    else trapezoid_tile(blockIdx.x - pre[seg], ntr, ntc, &ti, &tj);
    const int Rv = R - 2;                             // valid rows (rhs row included, pad rows excluded)
    double *Fg = pool + D_.off;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i0 = c0 + ti * TW + (wave & 1) * QW;      // row origin of this wave's quadrant
    const int j0 = c0 + tj * TW + (wave >> 1) * QW;     // col origin
    if (i0 >= Rv || j0 >= CH) return;
    if (i0 + QW - 1 < j0) return;                     // quadrant strictly above the diagonal
    const int l15 = lane & 15, l4 = lane >> 4;
    d4_t acc[Q][Q];
#pragma unroll
    for (int a = 0; a < Q; a++)
#pragma unroll
        for (int b = 0; b < Q; b++) acc[a][b] = (d4_t){ 0, 0, 0, 0 };
    // operand rows, clamped so that every load is in range; out-of-range rows/columns are discarded at the store
    int rj[Q], ri[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) { rj[q] = min(j0 + 16 * q + l15, Rv - 1); ri[q] = min(i0 + 16 * q + l15, Rv - 1); }
    const int nk = (wdt + 3) >> 2;
    // K in chunks of 8 steps (32 panel columns): the 32 operand loads of chunk c + 1 are issued before the 32 MFMAs of
    // chunk c (two register sets, ping-pong).  No branch around a load and no arithmetic on a loaded value before its
    // chunk's turn -- either makes hipcc wait for every outstanding load on the spot; the K tail is handled by clamping
    // the column (always a valid address) and zeroing the operand at the MFMA.
    auto load_chunk = [&](int c0, double (&v)[KCH][2 * Q]) {
#pragma unroll
        for (int u = 0; u < KCH; u++) {
            const double *col = Fg + (size_t)(k0 + min(4 * (c0 + u) + l4, wdt - 1)) * R;
#pragma unroll
            for (int q = 0; q < Q; q++) { v[u][q] = col[rj[q]]; v[u][Q + q] = col[ri[q]]; }
        }
    };
    auto mma_chunk = [&](int c0, const double (&v)[KCH][2 * Q]) {
#pragma unroll
        for (int u = 0; u < KCH; u++) {
            const bool kok = 4 * (c0 + u) + l4 < wdt;          // K tail (and steps past nk): zeros
            double pj[Q], pi[Q];
#pragma unroll
            for (int q = 0; q < Q; q++) { pj[q] = kok ? v[u][q] : 0.0; pi[q] = kok ? v[u][Q + q] : 0.0; }
#pragma unroll
            for (int a = 0; a < Q; a++)
#pragma unroll
                for (int b = 0; b < Q; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(pj[a], pi[b], acc[a][b], 0, 0, 0);
        }
    };
    // the 16 elements of C this lane updates are requested up front as well (clamped addresses, masked at the store)
    // D[a][b] element (row = l4 + 4*reg, col = l15) = update of C[i = i0+16b+l15, j = j0+16a+l4+4*reg]
    double cv[Q][Q][4];
    auto load_c = [&]() {
#pragma unroll
        for (int a = 0; a < Q; a++)
#pragma unroll
            for (int b = 0; b < Q; b++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int i = min(i0 + 16 * b + l15, Rv - 1), j = min(j0 + 16 * a + l4 + 4 * reg, CH - 1);
                    cv[a][b][reg] = Fg[(size_t)j * R + i];
                }
    };
    if constexpr (CV_EARLY) load_c();
    {
        double A[KCH][2 * Q], B[KCH][2 * Q];
        load_chunk(0, A);
        for (int c0 = 0; c0 < nk; c0 += 2 * KCH) {
            const bool hb = c0 + KCH < nk;
            if (hb) load_chunk(c0 + KCH, B);
            mma_chunk(c0, A);
            if (hb) {
                if (c0 + 2 * KCH < nk) load_chunk(c0 + 2 * KCH, A);
                mma_chunk(c0 + KCH, B);
            }
        }
    }
    if constexpr (!CV_EARLY) load_c();
#pragma unroll
    for (int a = 0; a < Q; a++)
#pragma unroll
        for (int b = 0; b < Q; b++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int i = i0 + 16 * b + l15, j = j0 + 16 * a + l4 + 4 * reg;
                if (i < Rv && j < CH && i >= j) Fg[(size_t)j * R + i] = cv[a][b][reg] - acc[a][b][reg];
            }
}
// Default: 2 K-steps of operands in flight per buffer and the C elements read after the K loop -- 92 registers, FIVE waves per SIMD.
// Round 4: the kernel is not short of bytes (an XCD-aware tile order that halves its HBM traffic moved it 1.4 %) but of waves to
// cover a tile's start (segment search -> front record -> first operands: dependent round trips) and end (C read-modify-write):
// with two 222-register waves per SIMD the matrix cores were busy 40 % of the time.  17.2 ms (8 K-steps, C up front, 2 waves) ->
// 15.9 (4 K-steps, 3 waves) -> 15.1 (C after the loop, 4 waves) -> 14.8 (2 K-steps, 5 waves) -> 19.4 (6 waves: spills) per
// iteration of the 1 M lattice = 32 TFLOP/s by its own flops, against 48.6 TFLOP/s that back-to-back v_mfma_f64_16x16x4_f64
// reach on this part at ANY occupancy and accumulator count (tools/ubench/mfma_f64.hip; the data sheet says 78.6).
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(5, 5))) k_syrk_big(DevPlan P, const int *__restrict__ list, const int *__restrict__ pre, int n,
                                                  int s_lo, int s_hi, int mode, double *__restrict__ pool) {
    syrk_big_body<2, false>(P, list, pre, n, s_lo, s_hi, mode, pool);
}
// Launches with few tiles (every wide update of the 100 k lattice: at most 300 tiles of 64 x 64 for 1 280 workgroup slots): 32 x 32 tiles,
// one 16 x 16 MFMA block per wave -- four times the workgroups, a quarter of the K loop each; the launch lasts as long as its slowest tile
__global__ void __launch_bounds__(TPB) k_syrk_big32(DevPlan P, const int *__restrict__ list, const int *__restrict__ pre, int n,
                                                    int s_lo, int s_hi, int mode, double *__restrict__ pool) {
    syrk_big_body<4, false, 1>(P, list, pre, n, s_lo, s_hi, mode, pool);
}

// tile l of the column-major lower trapezoid: column tj holds (ntr - tj) tiles, tj < ntc
__device__ __forceinline__ void trapezoid_tile(int l, int ntr, int ntc, int *ti, int *tj_) {
    // float estimate (one v_sqrt_f32: the f64 sqrt expands to ~40 dependent instructions on the latency path of every
    // LDS-resident update), corrected exactly by the two integer loops below
    const float b = 2.0f * ntr + 1.0f;
    int tj = (int)((b - sqrtf(fmaxf(b * b - 8.0f * l, 0.0f))) * 0.5f);
    if (tj < 0) tj = 0;
    if (tj > ntc - 1) tj = ntc - 1;
    while (tj > 0 && tj * ntr - tj * (tj - 1) / 2 > l) tj--;
    while (tj + 1 < ntc && (tj + 1) * ntr - (tj + 1) * tj / 2 <= l) tj++;
    *tj_ = tj;
    *ti = tj + (l - (tj * ntr - tj * (tj - 1) / 2));
}

// ------------------------------------------------------------------------------------------------------
// backward substitution, one workgroup per front, levels from the root down.
//   x_T = L11^-T ( y_T - L21^T x_struct )      (row-dot form of smatd_utriangle_solve, smatd.c:1075)
// xw (LDS, R doubles) holds x over the front's rows: struct part gathered from the global x, own part
// filled as it is solved, NB columns at a time from the last block to the first.
// ------------------------------------------------------------------------------------------------------
// wave-wide sum through the DPP cross-lane paths of the VALU (no LDS crossbar traffic: six ds_bpermute round trips
// per value were the longest phase of k_backsolve).  Fixed order; the result is valid in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double v) {      // lanes without a source (or in a masked row) receive 0
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum63(double v) {
    v += dpp_take<0x111, 0xf>(v);      // row_shr:1
    v += dpp_take<0x112, 0xf>(v);      // row_shr:2
    v += dpp_take<0x114, 0xf>(v);      // row_shr:4
    v += dpp_take<0x118, 0xf>(v);      // row_shr:8   -> lane 15 of every 16-lane row holds the row sum
    v += dpp_take<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v += dpp_take<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return v;
}

// Back substitution of fronts with a large update block, part 1: t_c = sum_{i >= ns} L[i][c] x[i] (the product with the
// ALREADY KNOWN x of the update rows) has no dependency inside the front, so it is spread over ceil(ns / 32) workgroups
// per front instead of being streamed by the single workgroup that owns the triangular solve.  t lands in x at the
// front's own positions, where k_backsolve (split = 1) picks it up before it overwrites them with the solution.
constexpr long long BS_SPLIT_MIN = 1 << 17;       // update rows x own columns from which the split pays
constexpr int BS_CHUNK = 4096;                    // update rows staged in LDS at a time
// (fronts wider than one outer block always: k_backsolve_blk takes their update-row product from this kernel)
__host__ __device__ inline bool bs_split_front(int nsb, int nub) { return nub > 0 && (9ll * nsb * nub >= BS_SPLIT_MIN || 3 * nsb > NB * 4); }
__global__ void __launch_bounds__(TPB) k_backsolve_gemv(DevPlan P, const int *__restrict__ list, const int *__restrict__ pre, int n,
                                                        const double *__restrict__ pool, double *__restrict__ x) {
    __shared__ double xu[BS_CHUNK];
    const int seg = find_seg(pre, n, blockIdx.x);
    const int t = list[seg], cb = blockIdx.x - pre[seg];
    const FrontDesc D_ = P.fd[t];
    const int nsb = D_.nsb, nub = D_.nub, nbc = nsb + nub;
    const int R = 3 * (nbc + 1), ns = 3 * nsb, nu = 3 * nub;
    const int k0 = cb * NB, wdt = min(NB, ns - k0);
    const double *Fg = pool + D_.off;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int *rows = P.f_rows + D_.rows_begin;
    const int c0 = wave * 8;
    double acc[8];
    const double *colp[8];
#pragma unroll
    for (int q = 0; q < 8; q++) { acc[q] = 0; colp[q] = Fg + (size_t)(k0 + min(c0 + q, wdt - 1)) * R + ns; }
    for (int r0 = 0; r0 < nu; r0 += BS_CHUNK) {
        const int nr = min(BS_CHUNK, nu - r0);
        __syncthreads();
        for (int e = tid; e < nr; e += TPB) xu[e] = x[(size_t)3 * rows[(r0 + e) / 3] + (r0 + e) % 3];
        __syncthreads();
        int i = lane;
        for (; i + 192 < nr; i += 256) {          // 32 loads per lane in flight; same summation order as the plain loop
            double v[4][8];
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int q = 0; q < 8; q++) v[u][q] = colp[q][r0 + i + 64 * u];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const double xv = xu[i + 64 * u];
#pragma unroll
                for (int q = 0; q < 8; q++) acc[q] = fma(v[u][q], xv, acc[q]);
            }
        }
        for (; i < nr; i += 64) {
            const double xv = xu[i];
#pragma unroll
            for (int q = 0; q < 8; q++) acc[q] = fma(colp[q][r0 + i], xv, acc[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const double sum = wave_sum63(acc[q]);
        if (lane == 63 && c0 + q < wdt) x[(size_t)3 * D_.first + k0 + c0 + q] = sum;
    }
}

// PRE (the multi-level launch): while the workgroup waits for its parent it copies its whole L panel into LDS, so that
// nothing on the critical path -- parent done -> x gathered -> column blocks -> publish -- touches HBM / L2 except the
// gather of a few x values.
__host__ __device__ inline size_t backsolve_lds(int m, int ns, bool pre) { return (size_t)(m + NB + 8 + NB * (NB + 1) + (pre ? (size_t)ns * ((m + 3) | 1) : 0)) * 8; }
// upd (batch path): the state update of the front's own poses rides along -- state = l_point + dx with theta wrap, NaN guard,
// optional pinned host mirrors (april_graph_xyt.c:302-314, see k_update_states) -- instead of a kernel of its own at the end
struct UpdArgs { const int *perm; const double *lp; double *st, *dX, *st_out, *dx_out; int *bad_out; double *lp_next; };      // lp_next (resident loops): the new state is also the next step's linearisation point (aprilsam.c:131-135) -- no copy between two steps
// common tail of the back-substitution kernels: x of the own columns to HBM, flag for the children, then the state update
// of the own poses (april_graph_xyt.c:302-314)
__device__ __forceinline__ void backsolve_finish(const FrontDesc &D_, int t, const double *xw, double *__restrict__ x, int *xflags, int *bad, const UpdArgs &upd, int ev = 0) {
    const int tid = threadIdx.x, nsb = D_.nsb, ns = 3 * nsb;
    for (int e = tid; e < ns; e += TPB) x[(size_t)3 * D_.first + e] = xw[e];
    if (xflags) publish_flag(xflags + t, ev);       // the children may go; the state update of the own poses follows
    if (upd.perm) {
        for (int k = tid; k < nsb; k += TPB) {
            const int i = upd.perm[D_.first + k];
            const double d0 = xw[3 * k], d1 = xw[3 * k + 1], d2 = xw[3 * k + 2];
            if (isnan(d0) || isnan(d1) || isnan(d2)) {
                const double qnan = __longlong_as_double(0x7ff8000000000000ll);
                upd.dX[3 * i + 0] = qnan; upd.dX[3 * i + 1] = qnan; upd.dX[3 * i + 2] = qnan;
                if (upd.dx_out) { upd.dx_out[3 * i + 0] = qnan; upd.dx_out[3 * i + 1] = qnan; upd.dx_out[3 * i + 2] = qnan; }
                continue;
            }
            const double s0 = upd.lp[3 * i + 0] + d0, s1 = upd.lp[3 * i + 1] + d1, s2 = mod2pi_dev(upd.lp[3 * i + 2] + d2);
            if (upd.st) { upd.st[3 * i + 0] = s0; upd.st[3 * i + 1] = s1; upd.st[3 * i + 2] = s2; }
            if (upd.lp_next) { upd.lp_next[3 * i + 0] = s0; upd.lp_next[3 * i + 1] = s1; upd.lp_next[3 * i + 2] = s2; }      // (a skipped pose keeps state = l_point: nothing to write)
            upd.dX[3 * i + 0] = d0; upd.dX[3 * i + 1] = d1; upd.dX[3 * i + 2] = d2;
            if (upd.st_out) {
                upd.st_out[3 * i + 0] = s0; upd.st_out[3 * i + 1] = s1; upd.st_out[3 * i + 2] = s2;
                upd.dx_out[3 * i + 0] = d0; upd.dx_out[3 * i + 1] = d1; upd.dx_out[3 * i + 2] = d2;
            }
        }
    }
    // failure record -> pinned mirror.  The host zeroes the mirror before the launch; a workgroup only ever writes a SET
    // record, and every workgroup of the last launch looks (the one that raises a flag late -- a dependency time-out in this
    // very launch -- sees its own atomicCAS): nothing is lost to the order in which workgroups finish.
    if (upd.bad_out && tid == 0 && __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) upd.bad_out[k] = __hip_atomic_load(bad + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// TALL (levels with fronts of >= BS_TALL_ROWS rows, chosen by the host per launch): the product loop keeps 32 loads per lane
// in flight instead of 8 -- a single workgroup streams the whole L11 triangle of a big front -- at the price of VGPRs
// (2 workgroups per CU instead of 3, which the wide levels of small fronts would feel)
constexpr int BS_TALL_ROWS = 768;
template <bool PRE, bool TALL = false>
__global__ void __launch_bounds__(TPB) k_backsolve_t(DevPlan P, const int *__restrict__ fronts, const double *__restrict__ pool,
                                                     double *__restrict__ x, int split, int *xflags, int wait, int *bad, UpdArgs upd) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int t = fronts[blockIdx.x];
    const int ev = xflags ? flag_value(P) : 0;
    const FrontDesc D_ = P.fd[t];
    const int nsb = D_.nsb, nub = D_.nub, nbc = nsb + nub;
    const int R = 3 * (nbc + 1), ns = 3 * nsb, m = 3 * nbc;
    double *xw = smem;                 // m doubles: x over the front's rows
    double *part = smem + m;           // NB doubles: right-hand side of the current column block
    double *Ldg = smem + m + NB + 8;   // NB x (NB+1): diagonal block of the current step, column c at Ldg[c*(NB+1)]
    const double *Fg = pool + D_.off;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int *rows = P.f_rows + D_.rows_begin;
    long long *pf = (P.prof && P.prof_mode == 2) ? P.prof + (size_t)t * PROF_SLOTS : nullptr;
    double *Lp = smem + m + NB + 8 + NB * (NB + 1);      // PRE: the L panel, column c at Lp[c * ldp]
    const int ldp = R | 1;
    auto Lat = [&](int c, int r) -> double { if constexpr (PRE) return Lp[c * ldp + r]; else return Fg[(size_t)c * R + r]; };
    if constexpr (PRE) {
        // (the own columns are contiguous in the front: element e; eight loads per thread in flight)
        const int ne = ns * R;
        for (int e0 = tid; e0 < ne; e0 += 8 * TPB) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = Fg[min(e0 + u * TPB, ne - 1)];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int e = e0 + u * TPB; if (e < ne) { const int c = e / R; Lp[c * ldp + (e - c * R)] = v[u]; } }
        }
    }
    if (wait && D_.parent >= 0) {          // multi-level launch, root first: x of every ancestor is final once the parent is done
        if (tid == 0) wait_flag(xflags + D_.parent, bad, ev);
        if constexpr (PRE) { __syncthreads(); asm volatile("" ::: "memory"); }     // no acquire fence: x is gathered with ld_agent below
        else acquire_all();
    }
    if (pf && tid == 0) pf[4] = wall_clock64();
    // split: the update-row product of this front was computed by k_backsolve_gemv and waits in x at the own positions
    const bool pre_t = split && bs_split_front(nsb, nub);
    const int mprod = pre_t ? ns : m;                  // rows that still enter the products below
    // the own part of xw starts as the right-hand side y (row m of the own columns; minus the update-row product when
    // k_backsolve_gemv left it in x) and turns into the solution block by block
    if (pre_t) { for (int e = tid; e < ns; e += TPB) xw[e] = Fg[(size_t)e * R + m] - x[(size_t)3 * D_.first + e]; }
    else {
        for (int e = tid; e < ns; e += TPB) xw[e] = Fg[(size_t)e * R + m];
        for (int e = tid; e < 3 * nub; e += TPB) { const double *xp = x + (size_t)3 * rows[e / 3] + e % 3; if constexpr (PRE) xw[ns + e] = ld_agent(xp); else xw[ns + e] = *xp; }
    }
    __syncthreads();
    if (pf && tid == 0) pf[5] = wall_clock64();
    long long t_prod = 0, t_sum = 0, t_solve = 0, ts = 0;
    // TALL: L does not depend on x, so the loads of a step -- its diagonal block and the first 512 rows of its product --
    // are issued during the PREVIOUS step, right after that step's own fmas, and fly while the sums, the barriers and
    // the in-block solve run; a step then costs fma issue instead of a memory round trip.  Rows past the end are
    // clamped and enter with x = 0; every accumulator still sees its rows in the same order.
    constexpr int PFU = 8;                       // prefetched chunks of 64 rows
    double pv[TALL ? PFU : 1][8], pdg[4];
    auto prefetch = [&](int k1p) {
        if constexpr (TALL && !PRE) {
            const int k0p = max(0, k1p - NB), wdtp = k1p - k0p;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int e = tid + q * TPB, r = e & 31, cc = e >> 5;
                pdg[q] = Fg[(size_t)(k0p + min(cc, wdtp - 1)) * R + k0p + min(r, wdtp - 1)];
            }
#pragma unroll
            for (int u = 0; u < PFU; u++) {
                const int row = min(k1p + lane + 64 * u, mprod - 1);
                if (k1p + 64 * u < mprod) {             // (workgroup-uniform) a chunk past the end costs no load instructions
#pragma unroll
                    for (int q = 0; q < 8; q++) pv[u][q] = Fg[(size_t)(k0p + min(wave * 8 + q, wdtp - 1)) * R + row];
                } else {
#pragma unroll
                    for (int q = 0; q < 8; q++) pv[u][q] = 0.0;
                }
            }
        }
    };
    prefetch(ns);
    for (int k1 = ns; k1 > 0; k1 -= NB) {
        const int k0 = max(0, k1 - NB), wdt = k1 - k0;
        if (pf) ts = wall_clock64();
        // the diagonal block of this step is requested first, by everybody and coalesced (4 values per thread); it is
        // parked in LDS after the products so that both sets of loads share one memory round trip
        double dg[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int e = tid + q * TPB, r = e & 31, cc = e >> 5;
            if constexpr (TALL && !PRE) dg[q] = pdg[q]; else dg[q] = Lat(k0 + min(cc, wdt - 1), k0 + min(r, wdt - 1));
        }
        // w_c = y_c - sum_{i >= k1} L[i,c] x[i]: each wave owns 8 of the (<= 32) columns and streams them
        // together (8 independent loads in flight per lane), lanes run down the rows (coalesced)
        {
            const int c0 = wave * 8;
            double acc[8];
            int colc[8];
#pragma unroll
            for (int q = 0; q < 8; q++) { acc[q] = 0; colc[q] = k0 + min(c0 + q, wdt - 1); }
            int i = k1 + lane;
            if constexpr (TALL && !PRE) {
                {
                    double xv[PFU];
#pragma unroll
                    for (int u = 0; u < PFU; u++) xv[u] = (i + 64 * u < mprod) ? xw[min(i + 64 * u, mprod - 1)] : 0.0;
#pragma unroll
                    for (int u = 0; u < PFU; u++)
#pragma unroll
                        for (int q = 0; q < 8; q++) acc[q] = fma(pv[u][q], xv[u], acc[q]);
                    i += 64 * PFU;
                }
                while (i - lane < mprod) {           // fronts with more than 512 product rows: the rest in batches of 4 chunks
                    double v[4][8], xv[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int row = min(i + 64 * u, mprod - 1);
#pragma unroll
                        for (int q = 0; q < 8; q++) v[u][q] = Fg[(size_t)colc[q] * R + row];
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) xv[u] = (i + 64 * u < mprod) ? xw[min(i + 64 * u, mprod - 1)] : 0.0;
#pragma unroll
                    for (int u = 0; u < 4; u++)
#pragma unroll
                        for (int q = 0; q < 8; q++) acc[q] = fma(v[u][q], xv[u], acc[q]);
                    i += 256;
                }
                if (k0 > 0) prefetch(k0);
            }
            for (; i < mprod; i += 64) {
                const double xv = xw[i];
#pragma unroll
                for (int q = 0; q < 8; q++) acc[q] = fma(Lat(colc[q], i), xv, acc[q]);
            }
            if (pf) { const long long tn = wall_clock64(); t_prod += tn - ts; ts = tn; }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const double sum = wave_sum63(acc[q]);
                if (lane == 63 && c0 + q < wdt) part[c0 + q] = xw[k0 + c0 + q] - sum;
            }
        }
        // parked for the in-block solve: strictly lower part (zero elsewhere, so that the chain below needs no predicate),
        // 1 / L[c][c] in the pad slot of column c -- the 32 divisions run here in parallel, not on wave 0's chain
        {
            const int r = tid & 31, dq = r - (tid >> 5);          // dq = 8 q for the one q (if any) whose element is L[r][r]
            double dsel = dg[0];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int cc = (tid >> 5) + 8 * q;
                Ldg[cc * (NB + 1) + r] = (r < wdt && r > cc) ? dg[q] : 0.0;      // (r > cc, r < wdt) implies cc < wdt
                dsel = (dq == 8 * q) ? dg[q] : dsel;
            }
            if (dq >= 0 && (dq & 7) == 0) Ldg[r * (NB + 1) + NB] = (r < wdt) ? 1.0 / dsel : 1.0;
        }
        if constexpr (TALL && !PRE) lds_barrier(); else __syncthreads();
        if (pf) { const long long tn = wall_clock64(); t_sum += tn - ts; ts = tn; }
        if (pf && tid == 0 && k1 == ns) pf[6] = wall_clock64();
        // in-block solve L[k0..k1)^T x = w on wave 0: lane c keeps column c of the block in registers, scaled by
        // 1 / L[c][c] like its right-hand side, x_i is broadcast by v_readlane; the 31-step recurrence is readlane + fma,
        // no LDS traffic, no barrier, no predicate (entries at and above the diagonal are zero)
        if (wave == 0) {
            const int c = lane & 31;
            double Lc[NB];
#pragma unroll
            for (int i = 0; i < NB; i++) Lc[i] = Ldg[c * (NB + 1) + i];
            const double rinv = Ldg[c * (NB + 1) + NB];
            double w = ((c < wdt) ? part[c] : 0.0) * rinv;
#pragma unroll
            for (int i = 0; i < NB; i++) Lc[i] *= rinv;
#pragma unroll
            for (int i = NB - 1; i > 0; i--) {
                const double xi = readlane_d(w, i);          // lane i holds the finished x_i
                w = fma(-Lc[i], xi, w);
            }
            if (lane < wdt) xw[k0 + lane] = w;
        }
        if constexpr (TALL && !PRE) lds_barrier(); else __syncthreads();
        if (pf) t_solve += wall_clock64() - ts;
    }
    if (pf && tid == 0) { pf[7] = wall_clock64(); pf[8] = t_prod; pf[9] = t_sum; pf[10] = t_solve; }
    backsolve_finish(D_, t, xw, x, xflags, bad, upd, ev);
}

// ------------------------------------------------------------------------------------------------------
// Back substitution of WIDE fronts on several compute units (option blk_backsolve): k_backsolve_t streams the whole L11
// triangle of a front through ONE workgroup, 32 dependent column blocks per 1 000 columns (7 % / 14 % of the HBM rate on
// the lattices, 141 us for the root of the 100 k lattice).  Here a front is taken 128 columns at a time -- the outer blocks
// of its factorisation -- by one CHAIN workgroup and up to six HELPER workgroups:
//   * chain: w_B = y_B - (update-row product, k_backsolve_gemv) - (contributions of every block solved so far, from the
//     helpers); then the 128 columns in four steps of 32, each x_s = inv(L_ss)^T w_s with the inverse diagonal blocks the
//     factorisation left behind (k_block_chain: no substitution chain at all, a 32 x 32 matrix-vector product), the other
//     sub-blocks updated from LDS; x_B published with a flag;
//   * helper h owns the 128-column blocks C = h (mod H): for every solved block b > C it adds L[rows b][cols C]^T x_b to its
//     sums (lanes down the rows, coalesced; all loads of a product in flight together), and hands block C over when b = C + 1.
// Dependencies run both ways (chain -> helpers: x_b; helpers -> chain: the sums), so the workgroups of a front must be
// resident together: the host only takes this path when a level's chains + helpers are a few dozen workgroups (a fraction
// of the device even with other streams busy); the polls are bounded like every other flag wait.
// ------------------------------------------------------------------------------------------------------
constexpr int BSB_MAX_HELPERS = 6, BSB_MAXB = 64, BSB_FAR = BSB_MAXB * OBW;      // blocks / doubles of scratch per front of a launch
__host__ __device__ inline int bsb_blocks(int ns) { return (ns + OBW - 1) / OBW; }
// blocks 0 .. nB - 3 receive contributions from the helpers (the chain adds the one of the block it has just solved itself)
__host__ __device__ inline int bsb_helpers(int ns) { const int nb = bsb_blocks(ns); return nb <= 2 ? 0 : (nb - 2 < BSB_MAX_HELPERS ? nb - 2 : BSB_MAX_HELPERS); }
__host__ __device__ inline size_t bsb_lds(int ns) { return (size_t)(ns + OBW + 512 + 6 * NB * (NB + 1) + OBP * NB * (NB + 1) + 64) * 8; }
// sum_r L[r0 + r][c0 + 32 wave + q] xb[r] for the wave's 32 columns q, r < wb <= 128 (xb: 128 doubles in LDS, zero past wb): lanes
// down the rows (coalesced), all 64 loads of a lane in flight before the first use; the sums arrive in lane 63
struct Gemv128 { double v[32][2]; };
__device__ __forceinline__ void gemv128_load(Gemv128 &g, const double *__restrict__ Fg, int R, int r0, int wb, int c0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 32; q++)
#pragma unroll
        for (int u = 0; u < 2; u++) g.v[q][u] = Fg[(size_t)(c0 + 32 * wave + q) * R + r0 + min(lane + 64 * u, wb - 1)];
}
template <class F> __device__ __forceinline__ void gemv128_sums(const Gemv128 &g, const double *__restrict__ xb, F &&out) {
    const int lane = threadIdx.x & 63;
    const double x0 = xb[lane], x1 = xb[lane + 64];
#pragma unroll
    for (int q = 0; q < 32; q++) {
        const double sum = wave_sum63(fma(g.v[q][1], x1, g.v[q][0] * x0));
        if (lane == 63) out(q, sum);
    }
}
__global__ void __launch_bounds__(TPB) k_backsolve_blk(DevPlan P, const int *__restrict__ list, const int *__restrict__ pre, int n,
                                                       const double *__restrict__ pool, double *__restrict__ x, const double *__restrict__ dinv_keep,
                                                       int *__restrict__ flags, double *__restrict__ far, int split, int *bad, UpdArgs upd) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int seg = find_seg(pre, n, blockIdx.x);
    const int t = list[seg], role = blockIdx.x - pre[seg];            // 0: chain, 1 .. H: helpers
    const FrontDesc D_ = P.fd[t];
    const int nsb = D_.nsb, nub = D_.nub, nbc = nsb + nub;
    const int R = 3 * (nbc + 1), ns = 3 * nsb, m = 3 * nbc;
    const int nB = bsb_blocks(ns), H = bsb_helpers(ns);
    const double *Fg = pool + D_.off;
    int *xflag = flags + (size_t)seg * 2 * BSB_MAXB, *tflag = xflag + BSB_MAXB;
    double *farf = far + (size_t)seg * BSB_FAR;
    const int tid = threadIdx.x, wave = tid >> 6;
    if (role > 0) {
        // ---- helper h: blocks C = h, h + H, ... <= nB - 3; for every solved block b >= C + 2 the product L[rows b][cols C]^T x_b ----
        const int h = role - 1;
        double *xb = smem;                          // x of the block just solved (<= 128)
        double *accL = smem + OBW;                  // sums of the owned blocks: accL[(C / H) * 128 + c]
        const int nown = (nB - 2 - h + H - 1) / H;
        for (int e = tid; e < nown * OBW; e += TPB) accL[e] = 0.0;
        for (int b = nB - 1; b >= 2; b--) {
            if (h > b - 2) break;                   // no owned block at or below b - 2 any more
            const int r0 = b * OBW, wb = min(OBW, ns - r0);
            int Ctop = b - 2; Ctop -= ((Ctop - h) % H + H) % H;          // largest owned C <= b - 2: the nearest one first
            Gemv128 g;
            gemv128_load(g, Fg, R, r0, wb, Ctop * OBW);                  // L does not depend on x: in flight while the flag is awaited
            if (tid == 0) wait_flag(xflag + b, bad);
            __syncthreads();
            for (int e = tid; e < OBW; e += TPB) xb[e] = e < wb ? ld_agent(x + (size_t)3 * D_.first + r0 + e) : 0.0;
            __syncthreads();
            for (int C = Ctop; C >= 0; C -= H) {
                if (C != Ctop) gemv128_load(g, Fg, R, r0, wb, C * OBW);
                gemv128_sums(g, xb, [&](int q, double sum) { accL[(C / H) * OBW + 32 * wave + q] += sum; });
                if (C == b - 2) {                   // every block above C + 1 has now contributed: hand it over
                    __syncthreads();
                    for (int e = tid; e < OBW; e += TPB) farf[(size_t)C * OBW + e] = accL[(C / H) * OBW + e];
                    publish_flag(tflag + C);
                }
            }
        }
        return;
    }
    // ---- chain -----------------------------------------------------------------------------------------------------------
    double *xw = smem;                              // ns: the solution of the own part (backsolve_finish reads it)
    double *w = smem + ns;                          // 128: right-hand side of the current block, turning into x
    double *xs = w + OBW;                           // 512: partial sums
    double *Ls = xs + 512;                          // six sub-diagonal 32 x 32 blocks of the current diagonal block: Ls[blk][i * 33 + j] = L[32 sr + i][32 sc + j]
    double *Iv = Ls + 6 * NB * (NB + 1);            // four inverse blocks: Iv[s][j * 33 + i] = inv(L_ss)[j][i]
    const bool pre_t = split && bs_split_front(nsb, nub);
    // the update-row product comes from k_backsolve_gemv (in x at the own positions); the host only sends fronts here that
    // have no update rows at all (a root) or enough of them for that kernel (bs_split_front)
    for (int e = tid; e < ns; e += TPB) xw[e] = pre_t ? x[(size_t)3 * D_.first + e] : 0.0;
    __syncthreads();
    for (int B = nB - 1; B >= 0; B--) {
        const int k0 = B * OBW, Wb = min(OBW, ns - k0), nsub = (Wb + NB - 1) / NB;
        // this block's pieces of L, of its inverses, and the block of L below it (the product with the block just solved): all
        // requested first, they fly while the helpers are waited for
        double lv[24], iv[16];
#pragma unroll
        for (int k = 0; k < 24; k++) {
            const int e = tid + TPB * k, bq = e >> 10, i = e & 31, j = (e >> 5) & 31;
            const int sr = bq == 0 ? 1 : (bq < 3 ? 2 : 3), sc = bq == 0 ? 0 : (bq < 3 ? bq - 1 : bq - 3);
            lv[k] = Fg[(size_t)(k0 + min(NB * sc + j, Wb - 1)) * R + k0 + min(NB * sr + i, Wb - 1)];      // (a partial last block of a root front: the columns past it lie past the end of the front -- and of the pool)
        }
        const double *dv = dinv_keep + ((size_t)D_.dinv0 + B * OBP) * (NB * NB);
#pragma unroll
        for (int k = 0; k < 16; k++) iv[k] = dv[tid + TPB * k];
        Gemv128 g;
        const int wnext = B < nB - 1 ? min(OBW, ns - (k0 + OBW)) : 1;
        if (B < nB - 1) gemv128_load(g, Fg, R, k0 + OBW, wnext, k0);
        if (B < nB - 2 && tid == 0) wait_flag(tflag + B, bad);
#pragma unroll
        for (int k = 0; k < 24; k++) {
            const int e = tid + TPB * k, bq = e >> 10, i = e & 31, j = (e >> 5) & 31;
            const int sr = bq == 0 ? 1 : (bq < 3 ? 2 : 3);
            Ls[bq * (NB * (NB + 1)) + i * (NB + 1) + j] = (NB * sr + i < Wb) ? lv[k] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 16; k++) { const int e = tid + TPB * k; Iv[(e >> 10) * (NB * (NB + 1)) + ((e >> 5) & 31) * (NB + 1) + (e & 31)] = iv[k]; }
        // w still holds x of the block just solved (zero past its width): its product with the block below the diagonal block
        if (B < nB - 1) gemv128_sums(g, w, [&](int q, double sum) { xs[32 * wave + q] = sum; });
        __syncthreads();                            // (also: the helpers' sums are visible -- read with ld_agent below)
        {
            double v = 0.0;
            const int e = tid;
            if (e < Wb) {
                v = Fg[(size_t)(k0 + e) * R + m] - xw[k0 + e];
                if (B < nB - 1) v -= xs[e];
                if (B < nB - 2) v -= ld_agent(farf + (size_t)B * OBW + e);
            }
            __syncthreads();
            if (e < OBW) w[e] = v;
        }
        __syncthreads();
        for (int sb = nsub - 1; sb >= 0; sb--) {
            // x_s = inv(L_ss)^T w_s: thread (i, part): eight partial sums per component, then one pass to add them up
            {
                const int i = tid & 31, part = tid >> 5;
                double acc = 0;
#pragma unroll
                for (int jj = 0; jj < 4; jj++) { const int j = 4 * part + jj; acc = fma(Iv[sb * (NB * (NB + 1)) + j * (NB + 1) + i], w[NB * sb + j], acc); }
                xs[part * NB + i] = acc;
            }
            __syncthreads();
            if (tid < NB) {
                double v = 0;
#pragma unroll
                for (int part = 0; part < 8; part++) v += xs[part * NB + tid];
                w[NB * sb + tid] = v;               // (the solved part replaces the right-hand side in place)
                if (NB * sb + tid < Wb) xw[k0 + NB * sb + tid] = v;
            }
            __syncthreads();
            // w_s' -= L[rows s][cols s']^T x_s for the sub-blocks left of s: thread (column j of s', half of the rows)
            if (sb > 0) {
                const int q = tid >> 6, l = tid & 63, j = l & 31, half = l >> 5;
                if (q < sb) {
                    const int bq = (sb == 1 ? 0 : (sb == 2 ? 1 : 3)) + q;             // block (sr = sb, sc = q) in the order staged above
                    double acc = 0;
#pragma unroll
                    for (int ii = 0; ii < 16; ii++) { const int i = 16 * half + ii; acc = fma(Ls[bq * (NB * (NB + 1)) + i * (NB + 1) + j], w[NB * sb + i], acc); }
                    xs[256 + (q * 2 + half) * NB + j] = acc;
                }
                __syncthreads();
                if (tid < NB * sb) { const int q = tid >> 5, j = tid & 31; w[NB * q + j] -= xs[256 + (q * 2) * NB + j] + xs[256 + (q * 2 + 1) * NB + j]; }
                __syncthreads();
            }
        }
        // publish x_B: the helpers fold it into everything two blocks and more to its left
        for (int e = tid; e < Wb; e += TPB) x[(size_t)3 * D_.first + k0 + e] = xw[k0 + e];
        if (B >= 2 && H > 0) publish_flag(xflag + B);
    }
    __syncthreads();
    // every helper has handed over all its blocks (each hand-over was waited for above), nobody polls any more: flags back to 0
    for (int e = tid; e < 2 * BSB_MAXB; e += TPB) reset_flag(xflag + e, 0);
    backsolve_finish(D_, t, xw, x, nullptr, bad, upd);
}

// ------------------------------------------------------------------------------------------------------
// Back substitution, column-per-lane form: the multi-level launch (xflags != null) and the per-level launches of
// latency-bound levels whose fronts are small enough for their L panel to sit in LDS (copied while the workgroup waits for
// its parent, or first thing).  For such fronts the per-block machinery of k_backsolve_t --
// 8 wave reductions, a parked diagonal block, two barriers per 32 columns, ~1.3 us of instruction issue per block -- is
// most of its time.  Here a lane owns a COLUMN: its sum over the rows below needs no reduction across lanes (the four
// waves take every fourth row and their partial sums meet in LDS), and the triangular solve of up to 64 columns is one
// chain on wave 0 -- v_readlane of the finished x_i, one fma per lane, the lane's L entries prefetched from its LDS
// column (conflict-free: odd leading dimension).  Two barriers per 64 columns.
// ------------------------------------------------------------------------------------------------------
constexpr int BSW = 64;
constexpr int BSW_MAX_NS = NB + 8 + NB * (NB + 1) - 256;     // own columns a front may have in this kernel (diagonal parked in LDS)
// BSWT: columns per chain step, 64 -- or 32 where the caller's register budget (a 1024-thread workgroup: 128 VGPRs) cannot hold 64 of them
template <int BSWT = BSW>
__device__ __forceinline__ void backsolve_w_body(const DevPlan &P, const int t, const double *__restrict__ pool, double *__restrict__ x, int *xflags, int *bad,
                                                 const UpdArgs &upd, double *smem) {
    const int ev = xflags ? flag_value(P) : 0;
    const FrontDesc D_ = P.fd[t];
    const int nsb = D_.nsb, nub = D_.nub, nbc = nsb + nub;
    const int R = 3 * (nbc + 1), ns = 3 * nsb, m = 3 * nbc;
    double *xw = smem;                                   // m doubles: x over the front's rows (own part: rhs, then solution)
    double *partial = smem + m;                          // 4 x 64 partial sums (inside the NB + 8 + NB * (NB + 1) doubles of backsolve_lds)
    double *dgs = smem + m + 256;                        // the diagonal of L (ns doubles, same region: the host checks ns <= BSW_MAX_NS)
    double *Lp = smem + m + NB + 8 + NB * (NB + 1);      // the L panel, column c at Lp[c * ldp]
    const int ldp = R | 1;
    const double *Fg = pool + D_.off;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int *rows = P.f_rows + D_.rows_begin;
    long long *pf = (P.prof && P.prof_mode == 2) ? P.prof + (size_t)t * PROF_SLOTS : nullptr;
    {
        const int ne = ns * R;                           // (the own columns are contiguous in the front: element e)
        for (int e0 = tid; e0 < ne; e0 += 8 * TPB) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = Fg[min(e0 + u * TPB, ne - 1)];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int e = e0 + u * TPB;
                if (e < ne) {       // the copy holds zeros at and above the diagonal (the chain below needs no mask), the diagonal goes to dgs
                    const int c = e / R, r = e - c * R;
                    Lp[c * ldp + r] = r > c ? v[u] : 0.0;
                    if (r == c) dgs[c] = v[u];
                }
            }
        }
    }
    if (xflags && D_.parent >= 0) {                      // multi-level launch: x of every ancestor is final once the parent is done
        if (tid == 0) wait_flag(xflags + D_.parent, bad, ev);
        __syncthreads();                                 // no acquire fence: x is gathered with ld_agent
        asm volatile("" ::: "memory");
       
    }
    if (pf && tid == 0) pf[4] = wall_clock64();
    for (int e = tid; e < ns; e += TPB) xw[e] = Fg[(size_t)e * R + m];
    for (int e = tid; e < 3 * nub; e += TPB) xw[ns + e] = ld_agent(x + (size_t)3 * rows[e / 3] + e % 3);
    __syncthreads();
    if (pf && tid == 0) pf[5] = wall_clock64();
    for (int k1 = ns; k1 > 0; k1 -= BSWT) {
        const int k0 = max(0, k1 - BSWT), wdt = k1 - k0;
        const bool on = lane < wdt;
        const int c = k0 + (on ? lane : 0);
        const double *Lc = Lp + (size_t)c * ldp;
        // wave 0 fetches the lane's entries of the diagonal block (scaled by 1 / L[c][c], zero at and above the diagonal:
        // no predicate on the chain) while waves 1-3 sum over the rows below the block, every third row each
        auto chain = [&](auto w_) {
            constexpr int W = decltype(w_)::value;
            double l[W];
#pragma unroll
            for (int i = 1; i < W; i++) l[i] = Lc[k0 + min(i, wdt - 1)];      // steps past the block meet x_i = 0 (any finite entry will do), lanes past it are never read
            const double rinv = on ? 1.0 / dgs[c] : 1.0;
#pragma unroll
            for (int i = 1; i < W; i++) l[i] *= rinv;
            __syncthreads();                                                    // partial sums of waves 1-3
            if (pf && tid == 0 && k1 == ns) pf[6] = wall_clock64();
            double w = on ? (xw[c] - ((partial[64 + lane] + partial[128 + lane]) + partial[192 + lane])) * rinv : 0.0;
#pragma unroll
            for (int i = W - 1; i > 0; i--) {
                const double xi = readlane_d(w, i);                            // lane i holds the finished x_i
                w = fma(-l[i], xi, w);
            }
            if (on) xw[c] = w;
        };
        if (wave == 0) {
            if (wdt <= 16) chain(std::integral_constant<int, 16>{});
            else if (BSWT <= 32 || wdt <= 32) chain(std::integral_constant<int, 32>{});
            else if constexpr (BSWT > 32) chain(std::integral_constant<int, 64>{});
        } else {
            double acc0 = 0, acc1 = 0;
            int i = k1 + wave - 1;
            for (; i + 3 < m; i += 6) { acc0 = fma(Lc[i], xw[i], acc0); acc1 = fma(Lc[i + 3], xw[i + 3], acc1); }
            if (i < m) acc0 = fma(Lc[i], xw[i], acc0);
            partial[wave * 64 + lane] = acc0 + acc1;
            __syncthreads();
        }
        __syncthreads();
    }
    if (pf && tid == 0) pf[7] = wall_clock64();
    backsolve_finish(D_, t, xw, x, xflags, bad, upd, ev);
}
__global__ void __launch_bounds__(TPB) k_backsolve_w(DevPlan P, const int *__restrict__ fronts, const double *__restrict__ pool,
                                                     double *__restrict__ x, int *xflags, int *bad, UpdArgs upd) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    backsolve_w_body(P, fronts[blockIdx.x], pool, x, xflags, bad, upd, smem);
}

// Multi-GPU exchange: the Schur update of a front whose parent lives on another rank, packed for the wire.  Only the
// part the parent's assembly reads travels: per update column j (>= ns) the rows from the top of its diagonal
// 3 x 3 block down to the right-hand-side row.  upd_packed_offset(j) = doubles in front of column j.
__host__ __device__ inline long long upd_packed_offset(int R, int ns, int j) {
    const long long Rv = R - 2, nb0 = ns / 3, b = j / 3, e = j - 3 * b;
    return 3 * ((b - nb0) * Rv - 3 * (nb0 + b - 1) * (b - nb0) / 2) + e * (Rv - 3 * b);
}
// dir 0: front -> buf, 1: buf -> front; one workgroup per update column
// Debug option pool_guard: guard bands of `len` doubles behind every frontal array (and so at the end of the pool).  mode 0 fills them
// with NaN (all bits set); mode 1, after a step, counts the words that no longer hold the pattern: count[0] += words, count[1] = a band
// that was hit.  A read that strays into a band and is USED turns results into NaN (the parity tests see it); a write is counted here.
__global__ void __launch_bounds__(TPB) k_guard(int mode, const long long *__restrict__ offs, int n, int len, double *__restrict__ pool, int *count) {
    const int b = blockIdx.x;
    if (b >= n) return;
    unsigned long long *g = (unsigned long long *)(pool + offs[b]);
    int bad = 0;
    for (int i = threadIdx.x; i < len; i += TPB) {
        if (mode == 0) g[i] = ~0ull;
        else if (g[i] != ~0ull) bad++;
    }
    if (bad) { atomicAdd(count, bad); count[1] = b; }
}
// Debug option pool_poison: before a step, everything a multi-level launch hands from one workgroup to another is filled with NaN -- the
// UPDATE block of every front the step (re)factorises (what its parent's extend-add reads) and x at its own positions (what its children's
// back substitution gathers).  A dependency wait that passes early then yields NaN / "not positive definite" with certainty instead of
// the previous step's numbers, which are the right ones whenever the previous step solved the same system (the mask that hid round 5's
// release defect from every test but the soak).  list == null: fronts 0 .. n-1.  what: bit 0 = update blocks, bit 1 = x.  skip_mode (an
// incremental step's update records, or null): entries with mode != 0 are UPDATED in place from their old factor, not re-assembled -- left alone.
__global__ void __launch_bounds__(TPB) k_poison(DevPlan P, const int *__restrict__ list, int n, const int *__restrict__ skip_mode, int skip_stride, int what,
                                                double *__restrict__ pool, double *__restrict__ x) {
    if ((int)blockIdx.x >= n) return;
    if (skip_mode && skip_mode[(size_t)blockIdx.x * skip_stride]) return;
    const int t = list ? list[blockIdx.x] : (int)blockIdx.x;
    const FrontDesc D = P.fd[t];
    const int nbc = D.nsb + D.nub, R = 3 * (nbc + 1), ns = 3 * D.nsb, C = 3 * nbc;
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    double *Fg = pool + D.off;
    if (what & 1)
        for (int c = ns + (threadIdx.x >> 6); c < C; c += TPB / 64)
            for (int r = 3 * (c / 3) + (threadIdx.x & 63); r < R - 2; r += 64) Fg[(size_t)c * R + r] = qnan;
    if (what & 2) for (int e = threadIdx.x; e < ns; e += TPB) x[(size_t)3 * D.first + e] = qnan;
}
__global__ void __launch_bounds__(TPB) k_pack_update(double *__restrict__ front, int R, int ns, double *__restrict__ buf, int dir) {
    const int j = ns + blockIdx.x, Rv = R - 2, r0 = j / 3 * 3;
    double *__restrict__ col = front + (size_t)j * R;
    double *__restrict__ pk = buf + upd_packed_offset(R, ns, j) - r0;
    if (dir == 0) { for (int r = r0 + threadIdx.x; r < Rv; r += TPB) pk[r] = col[r]; }
    else { for (int r = r0 + threadIdx.x; r < Rv; r += TPB) col[r] = pk[r]; }
}

// Incremental steps change a dozen small device tables (descriptors of the regenerated fronts, their index lists, launch
// tables, slots of the new factors, the new factors themselves).  Instead of one copy-engine call per table (each a
// few microseconds of host time plus a blit kernel on the stream) the host stages all of them in ONE pinned buffer and
// this kernel scatters them: block b applies patch b, reading the payload across PCIe.
struct Patch { void *dst; long long src_off; long long bytes; };
__global__ void __launch_bounds__(TPB) k_apply_patches(const Patch *__restrict__ patches, const char *__restrict__ payload) {
    const Patch p = patches[blockIdx.x];
    const char *src = payload + p.src_off;
    char *dst = (char *)p.dst;
    if (((((size_t)dst) | ((size_t)src) | (size_t)p.bytes) & 3) == 0) {
        for (long long i = threadIdx.x; i < p.bytes / 4; i += TPB) ((int *)dst)[i] = ((const int *)src)[i];
    } else {
        for (long long i = threadIdx.x; i < p.bytes; i += TPB) dst[i] = src[i];
    }
}
// The same scatter plus the linearisation of the step's new factors, for incremental steps: ONE workgroup -- a wave per patch
// (all of them in flight together: one PCIe round trip for the headers, one for the payloads), a workgroup barrier, then a
// thread per new factor.  What the barrier orders is this workgroup's own global stores against its own global loads: the
// compute unit's vector L1 is shared by its waves and written through, nothing here goes through the scalar cache.
// Saves two launches per step (about 4 us of host time and a few us of GPU idle time each).
// epoch: the step counter the prologue advances (every flag of the step's launches carries the new value); marks / up_list: the fronts the
// step's multi-level launch regenerates get marks[t] = the new value (assemble_front waits for exactly those children), or null
struct IncFlags { int *epoch; int *marks; const int *up_list; int n_up; };
constexpr int INL_PATCHES = 24, INL_BYTES = 2048;      // a small step's patches travel in the kernel arguments: no PCIe read on the kernel's path
// Third argument of k_inc_prologue / k_inc_one.  Never touched by name in device code (a by-value aggregate that is indexed
// dynamically gets copied to scratch, all 2.6 KB of it per lane): read through the kernel-argument segment pointer instead.
struct alignas(16) InlinePatches { Patch hdr[INL_PATCHES]; char pay[INL_BYTES]; };
struct IncPrologue {
    const Patch *patches; const char *payload; int n_patch, f_begin, f_end;
    int inl;                    // 1: headers and payload are in the kernel's InlinePatches argument

    const int *fa, *fb; const double *Z, *Wm, *lp, *st; const unsigned char *swp; const int *slot_blk, *slot_rhs; double *Hc; int *bad;
    long long *stamps;          // APRILSAM_AMD_INC_PROFILE=2: wall-clock stamps (100 MHz) of k_inc_one's phases, in pinned host memory (else null)
    int *done; int seq;         // k_inc_one: completion word in pinned host memory, written after everything else (the host spins on it)
};
struct IncArgsHead { IncPrologue a; IncFlags fl; InlinePatches inl; };       // layout of the first three kernel arguments (offsetof below)
constexpr int TAIL_MAXF = 8;    // new factors of a step whose values tail_refactor takes from the LDS mirror
__device__ __forceinline__ void inc_prologue_body(const IncPrologue &a, const IncFlags &fl, double *mirror = nullptr) {
    const Patch *patches = a.patches; const char *payload = a.payload; const int n_patch = a.n_patch, f_begin = a.f_begin, f_end = a.f_end;
    const int *__restrict__ fa = a.fa, *__restrict__ fb = a.fb; const double *__restrict__ Z = a.Z, *__restrict__ Wm = a.Wm, *__restrict__ lp = a.lp, *__restrict__ st = a.st;
    const unsigned char *__restrict__ swp = a.swp; const int *__restrict__ slot_blk = a.slot_blk, *__restrict__ slot_rhs = a.slot_rhs; double *__restrict__ Hc = a.Hc; int *__restrict__ bad = a.bad;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nthr = (int)blockDim.x;
    if (threadIdx.x < 4) bad[threadIdx.x] = 0;          // failure record of this step (the first kernel of the step to touch it)
    const char *kargs = (const char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(IncArgsHead, inl);
    if (a.inl) { patches = (const Patch *)(kargs + offsetof(InlinePatches, hdr)); payload = kargs + offsetof(InlinePatches, pay); }
    for (int q = wave; q < n_patch; q += nthr >> 6) {
        const Patch p = patches[q];
        const char *src = payload + p.src_off;
        char *dst = (char *)p.dst;
        if (((((size_t)dst) | ((size_t)src) | (size_t)p.bytes) & 3) == 0) {
            // eight PCIe reads per lane in flight (the loop as written would wait for each load before issuing the next)
            const int nw = (int)(p.bytes / 4);
            for (int i0 = lane; i0 < nw; i0 += 512) {
                int v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = ((const int *)src)[min(i0 + 64 * u, nw - 1)];
#pragma unroll
                for (int u = 0; u < 8; u++) if (i0 + 64 * u < nw) ((int *)dst)[i0 + 64 * u] = v[u];
            }
        } else {
            for (long long i = lane; i < p.bytes; i += 64) dst[i] = src[i];
        }
    }
    // dependency flags of the step's two multi-level launches: nothing is reset -- the step counter advances, a finished front publishes the
    // new number, and the fronts this step regenerates are marked with it (a parent waits for exactly those children; the factors of the
    // others are older and complete)
    // (no static LDS here: the callers' dynamic LDS goes up to the 160 KB limit.  Thread 0's atomic has RETURNED -- it is performed in the L2 --
    // before the barrier; everybody reads the new number from there afterwards)
    if (fl.epoch && threadIdx.x == 0) { const int old = atomicAdd(fl.epoch, 1); asm volatile("s_waitcnt vmcnt(0)" :: "v"(old) : "memory"); }
    __syncthreads();                                    // (also: the patches above carry the list read below)
    if (a.stamps && threadIdx.x == 0) a.stamps[1] = wall_clock64();
    if (fl.marks && (int)threadIdx.x < fl.n_up) {
        const int ev = __hip_atomic_load(fl.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = threadIdx.x; i < fl.n_up; i += nthr) reset_flag(fl.marks + fl.up_list[i], ev);
    }
    if (f_end - f_begin > TAIL_MAXF) mirror = nullptr;
    for (int g0 = 0; g0 < f_end - f_begin; g0 += nthr)
        linearise_factor<false>(g0 + (int)threadIdx.x, f_begin, f_end, nullptr, fa, fb, Z, Wm, lp, st, swp, slot_blk, slot_rhs, Hc, nullptr, nullptr, nullptr, mirror);
}
__global__ void __launch_bounds__(1024) k_inc_prologue(IncPrologue a, IncFlags fl, InlinePatches) { inc_prologue_body(a, fl); }

// The commonest incremental step of a pose-by-pose run: a new pose and its odometry factor.  Both live in the LAST tail front,
// whose own poses are eliminated in id order and whose array keeps its shape while it fills up (the host pads its structure
// with phantom rows up to the capacity, see inc_fast_step): the columns in front of the first pose a new factor touches are
// unchanged, and so is everything the children contribute.  Instead of assembling and factorising the whole front again, the
// state of the elimination in front of that pose is recovered from the factor itself -- S = L_t L_t^T, c = L_t y over the
// trailing own columns (L_t: their diagonal block of L, y: their entries of the right-hand-side row) -- the new factors'
// blocks are added, and the at most TAILK trailing columns are factorised by ONE wave in registers (chain_block).
// a_idx: first own block column a new factor touches; n_old: own poses before this step.
struct TailStep { int t, a_idx, n_old, nsb, nbc, first; long long off; int solve, s_idx; };      // solve: back substitution over the own blocks >= s_idx (<= a_idx) in the same launch      // nsb / nbc / first / off: the front's descriptor AFTER this step (from the host: the
                                                                                   // kernel then needs nothing the prologue's patches bring, and can fetch the old factor before them)
constexpr int TAILQ = 24, TAILK = TAILQ / 3, TAIL_LD = 33;      // trailing columns (8 poses), leading dimension of the LDS arrays (rows <= TAILQ + 1)
__host__ __device__ inline size_t tail_refactor_lds() { return (size_t)(2 * TAILQ * TAIL_LD + TAILQ * TAILQ + 33 * TAIL_MAXF) * 8; }
__device__ __forceinline__ double *tail_mirror(double *S) { return S + 2 * TAILQ * TAIL_LD + TAILQ * TAILQ; }      // written by the prologue's linearisation
constexpr int TAIL_PRE = (TAILQ * (TAILQ + 1) + 255) / 256;      // old-factor values per thread at the smallest workgroup
// issued at the very start of k_inc_one: the old trailing columns of L (rows of the trailing own poses, lower part) and their
// right-hand-side entries (row qo) travel while the prologue works; they are parked in LDS by tail_refactor
template <int NT>
__device__ __forceinline__ void tail_prefetch(const TailStep ts, const double *__restrict__ pool, double (&v)[TAIL_PRE]) {
    const int R = 3 * (ts.nbc + 1), m = 3 * ts.nbc, c0 = 3 * ts.a_idx, qo = 3 * (ts.n_old - ts.a_idx);
    const double *Fg = pool + ts.off;
#pragma unroll
    for (int u = 0; u < TAIL_PRE; u++) {
        const int e = (int)threadIdx.x + u * NT;
        const int c = e / (qo + 1), r = e - c * (qo + 1);
        v[u] = (e < qo * (qo + 1) && (r >= c)) ? (r < qo ? Fg[(size_t)(c0 + c) * R + c0 + r] : Fg[(size_t)(c0 + c) * R + m]) : 0.0;
    }
}
// ts.solve: the columns of the solve window in FRONT of the re-factorised ones do not change in this step (nor do their entries of
// the right-hand-side row): lane c of wave 0 fetches its column -- so[0] the diagonal entry, so[i] the entry i rows into the
// window, so[TAILQ] the right-hand-side entry -- while the prologue works
__device__ __forceinline__ void tail_prefetch_solve(const TailStep ts, const double *__restrict__ pool, double (&so)[TAILQ + 1]) {
    const int R = 3 * (ts.nbc + 1), m = 3 * ts.nbc, w0 = 3 * ts.s_idx, off = 3 * (ts.a_idx - ts.s_idx), qs = 3 * (ts.nsb - ts.s_idx);
    const int c = threadIdx.x;
    const bool mine = ts.solve && c < off;
    const double *col = pool + ts.off + (size_t)(w0 + (mine ? c : 0)) * R;
#pragma unroll
    for (int i = 1; i < TAILQ; i++) so[i] = (mine && i > c && i < qs) ? col[w0 + i] : 0.0;
    so[0] = mine ? col[w0 + c] : 1.0;
    so[TAILQ] = mine ? col[m] : 0.0;
}
template <int NT>
__device__ __forceinline__ void tail_refactor(const TailStep ts, double *__restrict__ pool, const IncPrologue &a, double *S, const double (&pre)[TAIL_PRE],
                                              double *__restrict__ xg, const UpdArgs &upd, const double (&so)[TAILQ + 1]) {
    const int tid = threadIdx.x;
    const int nsb = ts.nsb, nbc = ts.nbc, R = 3 * (nbc + 1), m = 3 * nbc;
    const int c0 = 3 * ts.a_idx, q = 3 * (nsb - ts.a_idx), qo = 3 * (ts.n_old - ts.a_idx);
    double *Fg = pool + ts.off;
    struct { int first; } D{ ts.first };
    double *M = S, *Lt = S + TAILQ * TAIL_LD;
    const int nf = min(a.f_end - a.f_begin, TAIL_MAXF);
    int nas[TAIL_MAXF], nbs[TAIL_MAXF];                  // poses of the new factors (all loads in flight together, behind step 1's)
#pragma unroll
    for (int k = 0; k < TAIL_MAXF; k++) { const int f = min(a.f_begin + k, a.f_end - 1); nas[k] = a.fa[f]; nbs[k] = a.fb[f]; }
    // 1. the old trailing columns of L and their right-hand-side entries (row qo), fetched by tail_prefetch
#pragma unroll
    for (int u = 0; u < TAIL_PRE; u++) {
        const int e = tid + u * NT;
        if (e < qo * (qo + 1)) { const int c = e / (qo + 1), r = e - c * (qo + 1); Lt[c * TAIL_LD + r] = pre[u]; }
    }
    for (int e = tid; e < TAILQ * TAIL_LD; e += NT) M[e] = 0.0;
    __syncthreads();
    // 2. S = L_t L_t^T (lower part), c = L_t y: column j, row i >= j (i == qo: the right-hand-side row, kept at row q of M)
    for (int e = tid; e < qo * (qo + 1); e += NT) {
        const int j = e / (qo + 1), i = e - j * (qo + 1);
        if (i < j) continue;
        double acc = 0;
        if (i < qo) { for (int k = 0; k <= j; k++) acc = fma(Lt[k * TAIL_LD + i], Lt[k * TAIL_LD + j], acc); M[j * TAIL_LD + i] = acc; }
        else { for (int k = 0; k <= j; k++) acc = fma(Lt[k * TAIL_LD + j], Lt[k * TAIL_LD + qo], acc); M[j * TAIL_LD + q] = acc; }
    }
    __syncthreads();
    // 3. the new factors (linearised by the prologue into their slots): the 33 values of a factor by 33 lanes of wave 0, factor
    //    after factor (a wave's LDS operations stay in order: two factors may add to the same entry)
    if (tid < 33) {
        const double *mir = tail_mirror(S);            // (the host sends at most TAIL_MAXF factors this way)
#pragma unroll
        for (int k = 0; k < TAIL_MAXF; k++) {
            const int na = nas[k], nb = nbs[k];
            if (k >= nf || na < 0) continue;
            const int la = na - D.first - ts.a_idx, lb = nb >= 0 ? nb - D.first - ts.a_idx : -1;
            const int e = tid;
            const double v = mir[33 * k + e];
            if (e < 9) { const int i = e / 3, j = e - 3 * i; if (i >= j) M[(3 * la + j) * TAIL_LD + 3 * la + i] += v; }
            else if (e < 12) M[(3 * la + e - 9) * TAIL_LD + q] += v;
            else if (lb >= 0) {
                if (e < 21) { const int k = e - 12, i = k / 3, j = k - 3 * i, hi = max(la, lb), lo = min(la, lb); M[(3 * lo + j) * TAIL_LD + 3 * hi + i] += v; }
                else if (e < 30) { const int k = e - 21, i = k / 3, j = k - 3 * i; if (i >= j) M[(3 * lb + j) * TAIL_LD + 3 * lb + i] += v; }
                else M[(3 * lb + e - 30) * TAIL_LD + q] += v;
            }
        }
    }
    __syncthreads();
    // 4. the q trailing columns and the right-hand-side row, one wave, in registers
    //    (the chain runs all the steps of its width whatever q is: a new pose and its odometry factor are 6 columns, not 24)
    auto chain = [&](auto w_) {
        constexpr int W = decltype(w_)::value;
        double Dd[W];
        chain_block<NT, W>(M, TAIL_LD, 0, q, q + 1, a.bad, Dd, 1);
        __syncthreads();
        if (tid < 64) chain_store_diag<W>(M, TAIL_LD, 0, q, Dd);
        __syncthreads();
    };
    if (q <= 8) chain(std::integral_constant<int, 8>{}); else if (q <= 16) chain(std::integral_constant<int, 16>{}); else chain(std::integral_constant<int, TAILQ>{});
    // 4b. (ts.solve: every pose the step's walk visits lies in this window) the back substitution right here: the front is the root
    //     of the assembly tree, so x of its trailing columns depends on nothing but the trailing triangle that sits in LDS --
    //     L_t^T x = y_t, lane c owns column c (scaled by 1 / L[c][c], zeros at and above the diagonal), the finished x_i travel by
    //     v_readlane -- and the state update of the window's poses (april_graph_xyt.c:302-314).  Replaces a back-substitution
    //     pass over the whole front (its 84-column panel copied to LDS) for a step that moves its last few poses.
    if (ts.solve && tid < 64) {
        const int c = tid, off = 3 * (ts.a_idx - ts.s_idx), qs = off + q;      // the solve window: `off` unchanged columns (fetched by tail_prefetch_solve), then the q re-factorised ones
        const bool on = c < qs, fresh = c >= off;
        const double *Mc = M + (size_t)((on && fresh) ? c - off : 0) * TAIL_LD;
        double l[TAILQ];
#pragma unroll
        for (int i = 1; i < TAILQ; i++) l[i] = fresh ? ((on && i > c && i < qs) ? Mc[i - off] : 0.0) : so[i];
        const double rinv = on ? 1.0 / (fresh ? Mc[c - off] : so[0]) : 0.0;
        double w = on ? (fresh ? Mc[q] : so[TAILQ]) * rinv : 0.0;
#pragma unroll
        for (int i = 1; i < TAILQ; i++) l[i] *= rinv;
#pragma unroll
        for (int i = TAILQ - 1; i > 0; i--) {
            const double xi = readlane_d(w, i);
            w = fma(-l[i], xi, w);
        }
        const int k = c / 3, comp = c - 3 * k;
        const double x0 = __shfl(w, 3 * k, 64), x1 = __shfl(w, 3 * k + 1, 64), x2 = __shfl(w, 3 * k + 2, 64);
        if (on) {
            const int node = ts.first + ts.s_idx + k;                  // (tail poses: elimination position = node id)
            xg[(size_t)3 * node + comp] = w;
            const size_t o = (size_t)3 * node + comp;
            if (isnan(x0) || isnan(x1) || isnan(x2)) {                   // april_graph_xyt.c:304-305
                const double qnan = __longlong_as_double(0x7ff8000000000000ll);
                upd.dX[o] = qnan;
                if (upd.dx_out) upd.dx_out[o] = qnan;
            } else {
                double sv = upd.lp[o] + w;
                if (comp == 2) sv = mod2pi_dev(sv);
                upd.dX[o] = w;
                if (upd.st) upd.st[o] = sv;
                if (upd.st_out) { upd.st_out[o] = sv; upd.dx_out[o] = w; }
            }
        }
        if (upd.bad_out && tid == 0 && __hip_atomic_load(a.bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
#pragma unroll
            for (int kk = 0; kk < 4; kk++) upd.bad_out[kk] = __hip_atomic_load(a.bad + kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // 5. the columns go back to the front: trailing rows, zeros in the phantom rows, the right-hand-side row
    for (int c = tid >> 6; c < q; c += NT / 64) {
        const int cg = c0 + c;
        for (int r = 3 * (cg / 3) + (tid & 63); r <= m; r += 64)
            Fg[(size_t)cg * R + r] = r < c0 + q ? M[c * TAIL_LD + r - c0] : (r == m ? M[c * TAIL_LD + q] : 0.0);
    }
}

// A whole small incremental step in ONE launch of ONE workgroup: the prologue above, the step's regenerated fronts one after the
// other (children first; a new pose with its odometry factor regenerates the last tail front alone, a loop closure a short
// root path; or the last tail front's trailing columns alone, tail_refactor), then the back substitution over the visited
// root path from the top, the state update riding on it (n_dn = 0: left to the launches that follow).  Every
// phase reads what the previous one wrote through this compute unit's own vector L1 (written through); the tables the
// prologue patched are read by scalar loads afterwards, hence the scalar-cache invalidate.  The back-substitution body is
// written for TPB threads: the other waves leave before it (s_barrier counts the surviving waves only).
template <int NT>
__global__ void __launch_bounds__(NT) k_inc_one(IncPrologue a, IncFlags fl, InlinePatches, DevPlan P, TailStep ts, const int *__restrict__ up_list, int n_up,
                                                const int *__restrict__ dn_list, int n_dn,
                                                double *__restrict__ pool, long long full_lds_limit, double *__restrict__ x, UpdArgs upd, UpdCtx uc = UpdCtx{}) {
    extern __shared__ __attribute__((aligned(16))) double S[];
    long long *stamps = a.stamps;
    if (stamps && threadIdx.x == 0) stamps[0] = wall_clock64();
    double pre[TAIL_PRE], so[TAILQ + 1];
    if (ts.t >= 0) {
        tail_prefetch<NT>(ts, pool, pre);
        // (the 1024-thread variant has 128 VGPRs per thread: no room for the solve window's columns -- the host does not ask it to solve)
        if constexpr (NT < 1024) tail_prefetch_solve(ts, pool, so);
        else { for (int i = 0; i <= TAILQ; i++) so[i] = 0.0; }
    }
    inc_prologue_body(a, fl, ts.t >= 0 ? tail_mirror(S) : nullptr);
    __syncthreads();
    __builtin_amdgcn_s_dcache_inv();
    if (stamps && threadIdx.x == 0) stamps[2] = wall_clock64();
    if (ts.t >= 0) { tail_refactor<NT>(ts, pool, a, S, pre, x, upd, so); __syncthreads(); }
    for (int i = 0; i < n_up; i++) {
        const int t = up_list[i];
        if (uc.recs && uc.recs[i].mode) {               // updated, not re-factorised (front after front in this workgroup: no flags)
            UpdCtx ul = uc; ul.wflags = nullptr;
            front_update_body<NT>(P, t, uc.recs[i], ul, pool, nullptr, a.bad, S);
            __syncthreads();
            continue;
        }
        if (stamps && i == n_up - 1) {                  // profile: the phase stamps of the last front (DevPlan::prof slots) behind the kernel's own
            DevPlan Pp = P; Pp.prof = stamps + 8 - (size_t)t * PROF_SLOTS; Pp.prof_mode = 1;
            front_small_body<NT>(Pp, t, pool, a.Hc, a.bad, full_lds_limit, nullptr, 0, S);
            if (threadIdx.x == 0) { const FrontDesc D = P.fd[t]; stamps[5] = D.nsb; stamps[6] = D.nub; stamps[7] = D.ch_end - D.ch_begin; }
        } else
            front_small_body<NT>(P, t, pool, a.Hc, a.bad, full_lds_limit, nullptr, 0, S);
        __syncthreads();
    }
    if (stamps && threadIdx.x == 0) stamps[3] = wall_clock64();
    if (threadIdx.x >= TPB) return;
    for (int i = 0; i < n_dn; i++) {
        backsolve_w_body<(NT >= 1024 ? 32 : BSW)>(P, dn_list[i], pool, x, nullptr, a.bad, upd, S);
        __syncthreads();
    }
    if (stamps && threadIdx.x == 0) stamps[4] = wall_clock64();
    if (a.done) {                                       // every thread's stores to the pinned mirrors are performed before the word goes out
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {        // (release written out, see publish_flag)
            asm volatile("buffer_wbl2 sc0 sc1\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __hip_atomic_store(a.done, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// states AND l_points from the pinned mirrors (incremental steps keep the nodes' l_points, aprilsam.c:377-576)
__global__ void __launch_bounds__(TPB) k_load_states_lp(int n3, const double *__restrict__ host_st, const double *__restrict__ host_lp,
                                                        double *__restrict__ st, double *__restrict__ lp, int *__restrict__ bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (bad && i < 4) bad[i] = 0;
    if (i < n3) { st[i] = host_st[i]; lp[i] = host_lp[i]; }
}

// multi-GPU gather: states / l_points / dx of the nodes this rank owns, zeros elsewhere (summed over the ranks afterwards)
__global__ void __launch_bounds__(TPB) k_mask_owned(int n3, const int *__restrict__ nown, int rank, const double *__restrict__ st,
                                                    const double *__restrict__ lp, const double *__restrict__ dx, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n3) return;
    const bool mine = nown[i / 3] == rank;
    out[i] = mine ? st[i] : 0.0; out[n3 + i] = mine ? lp[i] : 0.0; out[2 * (size_t)n3 + i] = mine ? dx[i] : 0.0;
}

// states handed over by the caller in pinned host memory -> HBM (read across PCIe by the kernel itself: no copy-engine
// launch on the latency path of an API call); a batch step re-linearises every node first (aprilsam.c:131-135)
__global__ void __launch_bounds__(TPB) k_load_states(int n3, const double *__restrict__ host_st, double *__restrict__ st, double *__restrict__ lp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) { const double v = host_st[i]; st[i] = v; lp[i] = v; }
}

// state = l_point + dx with theta wrap; NaN in dx leaves the node untouched (april_graph_xyt.c:302-314).
// st_out / dx_out / bad_out (all or none): mirrors in pinned host memory the API call reads after its stream sync.
__global__ void __launch_bounds__(TPB) k_update_states(int N, const int *__restrict__ pos, const double *__restrict__ x,
                                                       const double *__restrict__ lp, double *__restrict__ st,
                                                       double *__restrict__ dX, double *__restrict__ st_out = nullptr,
                                                       double *__restrict__ dx_out = nullptr, const int *__restrict__ bad = nullptr,
                                                       int *__restrict__ bad_out = nullptr) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (bad_out && i < 4) bad_out[i] = bad[i];
    if (i >= N) return;
    const double *d = x + (size_t)3 * pos[i];
    double d0 = d[0], d1 = d[1], d2 = d[2];
    if (isnan(d0) || isnan(d1) || isnan(d2)) {        // april_graph_xyt.c:304-305: the node is skipped; NaN in dX tells the host
        const double qnan = __longlong_as_double(0x7ff8000000000000ll);
        dX[3 * i + 0] = qnan; dX[3 * i + 1] = qnan; dX[3 * i + 2] = qnan;
        if (dx_out) { dx_out[3 * i + 0] = qnan; dx_out[3 * i + 1] = qnan; dx_out[3 * i + 2] = qnan; }
        return;
    }
    const double s0 = lp[3 * i + 0] + d0, s1 = lp[3 * i + 1] + d1, s2 = mod2pi_dev(lp[3 * i + 2] + d2);
    st[3 * i + 0] = s0; st[3 * i + 1] = s1; st[3 * i + 2] = s2;
    dX[3 * i + 0] = d0; dX[3 * i + 1] = d1; dX[3 * i + 2] = d2;
    if (st_out) {
        st_out[3 * i + 0] = s0; st_out[3 * i + 1] = s1; st_out[3 * i + 2] = s2;
        dx_out[3 * i + 0] = d0; dx_out[3 * i + 1] = d1; dx_out[3 * i + 2] = d2;
    }
}

}  // namespace asam
