// plan.h — host-side symbolic plan of the multifrontal block-sparse Cholesky (MI355X-native design).
//
// The reference orders pose nodes with a constrained greedy min-degree (aprilsam.c:999-1249), hands
// the scalar matrix to CSparse's up-looking Cholesky in natural order (aprilsam.c:233-234) and keeps a
// block elimination tree (aprilsam.c:613-657).  Results do not depend on the ordering beyond ~1e-10
// (SURVEY.md §6), so this build uses its own: nested dissection on the pose graph, every separator and
// every leaf domain becoming one dense SUPERNODE ("front").  All indices below are in units of pose
// blocks (3 scalar unknowns each, aprilsam.c:141-148: idx = 3*position).
#pragma once
#include <cstdint>
#include <functional>
#include <vector>

namespace asam {

struct Plan {
    int N = 0;                 // pose nodes
    int F = 0;                 // factors
    int leaf_nodes = 16;

    std::vector<int> perm;     // elimination position -> node id   (== param->ordering semantics)
    std::vector<int> pos;      // node id -> elimination position

    // ---- fronts, numbered in post-order (children before parents) ---------------------------------
    int nF = 0;
    std::vector<int> f_first;  // first own position; own blocks are positions [f_first, f_first+f_nsb)
    std::vector<int> f_nsb;    // # own (eliminated) blocks
    std::vector<int> f_nub;    // # update blocks (struct rows: positions > own, sorted ascending)
    std::vector<int> f_parent; // assembly-tree parent (front owning min struct position), -1 = root
    std::vector<int> f_level;  // 0 = no children; parent level = 1 + max(children)
    std::vector<int64_t> f_rows_ptr;  // CSR into f_rows / f_rel, size nF+1
    std::vector<int> f_rows;   // struct block positions of each front
    std::vector<int> f_rel;    // for each struct block: its block index inside the PARENT front's row list
                               //   (own blocks first: 0..nsb_p-1, then parent's struct blocks)
    std::vector<int> ch_ptr, ch_idx;   // children CSR (ascending front index)
    std::vector<int64_t> f_off;        // offset (doubles) of the frontal array in the HBM front pool
    int64_t pool_doubles = 0;

    // frontal array of front t: column-major, R = 3*(nsb+nub+1) rows (last block row = RHS row + 2 pad
    // rows), C = 3*(nsb+nub) columns, leading dimension R.  After factorisation columns [0,3*nsb) hold
    // L (L11 over L21 over the solved RHS row y), the trailing block holds the Schur update.
    inline int rows(int t) const { return 3 * (f_nsb[t] + f_nub[t] + 1); }
    inline int cols(int t) const { return 3 * (f_nsb[t] + f_nub[t]); }

    // ---- factor -> front assignment -------------------------------------------------------------------
    std::vector<int> fac_front;       // owner front (front of the earliest-eliminated node of the factor)
    std::vector<int> fac_la, fac_lb;  // local block index of node a / b in the owner front (lb=-1 unary)
    // gather lists: destination 3x3 blocks of every front and the factor contributions landing there.
    // block destinations (lower triangle incl. diagonal blocks):
    std::vector<int> bd_front_ptr;    // per front: range of block destinations, size nF+1
    std::vector<int> bd_row, bd_col;  // local block row / col (row >= col)
    std::vector<int> bd_src_ptr;      // CSR into bd_src, size nBD+1
    std::vector<int> bd_src;          // contribution id = 3*factor + {0: (a,a), 1: off-diagonal, 2: (b,b)}
    // rhs destinations:
    std::vector<int> rd_front_ptr;    // per front: range of rhs destinations, size nF+1
    std::vector<int> rd_col;          // local block col
    std::vector<int> rd_src_ptr;      // CSR into rd_src
    std::vector<int> rd_src;          // contribution id = 2*factor + {0: g_a, 1: g_b}
    std::vector<uint8_t> fac_swap;    // 1 if the off-diagonal block must be stored transposed (la < lb)
    // the same lists flattened for the device: one record per destination block of a front, sorted by
    // (front, block col, block row) with brow = -1 for the rhs row; contributions are stored by the
    // linearise kernel in SLOT order (= destination order), so a front's inputs are one contiguous stream.
    struct DestRec { int brow, bcol, src_begin, src_end; };
    std::vector<int> dest_front_ptr;  // per front: range of DestRec, size nF+1
    std::vector<DestRec> dest;
    std::vector<int> slot_blk;        // 3 per factor: slot of (a,a), off-diagonal, (b,b) block (-1 if absent)
    std::vector<int> slot_rhs;        // 2 per factor: slot of g_a, g_b
    int n_slots = 0;

    // ---- level schedule ---------------------------------------------------------------------------------
    int nLevels = 0;
    std::vector<int> lev_ptr, lev_fronts;   // fronts of each level

    // ---- statistics -----------------------------------------------------------------------------------------
    int max_rows = 0;          // max over fronts of 3*(nsb+nub)
    int64_t nnzL = 0;          // scalar nnz of L incl. diagonal (dense fronts)
    double flops = 0;          // sum_j c_j^2
};

// Build ordering + symbolic plan.  factor_nodes: 2 ints per factor (second = -1 for unary factors).
// xy: optional 2 doubles per node (used only as a hint for geometric bisection; any values are valid).
void build_plan(Plan &P, int N, int F, const int *factor_nodes, const double *xy, int leaf_nodes);
void build_gather_lists(Plan &P);     // fills bd_* / rd_* (only the tests' host-side emulator reads them: derived on demand)

// ---- pieces, exposed for tests --------------------------------------------------------------------------
struct NDTree {
    struct Node { std::vector<int> verts; std::vector<int> children; };
    std::vector<Node> nodes;
    std::vector<int> roots;
};
void nested_dissection(int N, const std::vector<int> &adj_ptr, const std::vector<int> &adj,
                       const double *xy, int leaf_nodes, NDTree &tree);

// ---- the planner's thread pool (ordering.cpp), for the other phases of a plan ------------------------------
// PlanSession: the pool's workers stay awake (polling) while one exists; for graphs below `min_nodes` nodes it does nothing and
// plan_parallel_for runs on the calling thread.  plan_parallel_for: body(begin, end) over disjoint chunks of [0, n), each of
// at least `grain` items; the chunks write disjoint outputs, so the result does not depend on the number of threads.
struct PlanSession { explicit PlanSession(int n_nodes, int min_nodes = 1024); ~PlanSession(); PlanSession(const PlanSession &) = delete; bool on = false; };
void plan_parallel_for(int n, int grain, const std::function<void(int, int)> &body);

}  // namespace asam
