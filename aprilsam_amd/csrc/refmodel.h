// refmodel.h — host-side model of the reference's incremental bookkeeping (see refmodel.cpp)
#pragma once
#include <cstdint>
#include <functional>
#include <vector>

namespace asam {

std::vector<int> ref_min_degree_order(int N, const std::vector<std::vector<int>> &adj);

struct RefModel {
    bool valid = false;
    int N = 0, F = 0;
    std::vector<std::vector<int>> adj;      // pose adjacency, sorted, no self loops
    std::vector<int> ord, pos;              // position -> node, node -> position (reference order, identity-extended)
    std::vector<int> parent;                // block elimination tree, node -> parent node or -1
    std::vector<std::vector<int>> kids;     // ... and its children lists (unordered), kept in step with `parent`
    std::vector<int> visit_stack;
    std::vector<unsigned char> changed;     // label_changed
    std::vector<unsigned char> relin;       // label_relinearized
    int start_over = 0, naffected = 0, root = -1;
    int old_old_cross = 0;                  // inc_begin: new factors between two OLD poses of different branches of the tree (see inc_begin)

    void add_factor_edges(int a, int b);
    void set_parent(int v, int p);
    void insert_edge(int u, int v);
    int check_tree() const;
    void batch(int n_nodes, int n_factors, const int *fa, const int *fb);
    void inc_begin(int n_nodes, int n_factors, const int *fa, const int *fb);
    struct Visit { int node; bool update; };
    void plan_visit(std::vector<Visit> &out);
    void count_relinearized(const double *x, double dxy, double dth, const std::vector<Visit> &visits);
    void solve_visit(const double *x, double dxy, double dth, const std::function<void(int, bool)> &visit);
};

}  // namespace asam
