// Structure-aware solve of the reduced camera system: device plan and launch interface (kernels: ba_chol_sparse.hip; host planning:
// ba_sparse_plan.cpp).
#pragma once

#include "ba_kernels.hpp"
#include "ba_sparse_plan.hpp"

namespace esfm {

struct SparseWg {            // one workgroup of chol_sparse_kernel (SparsePlan::Wg with upd0 / upd1 counted in pairs)
    int32_t I, J, slot, upd0, upd1, dslot, kind;
};

struct SparseDev {           // kernel argument: the plan's tables and the solve's work areas, all device memory
    int nb = 0, n_tiles = 0, n_wgs = 0, n_blocks = 0;
    int32_t *col_src = nullptr;      // [nb * 64] original unknown of a permuted column, -1: padding
    int32_t *tile_ij = nullptr;      // [2 n_tiles] block row / column of a slot
    SparseWg *wgs = nullptr;         // [n_wgs] in dispatch order
    int32_t *upd = nullptr;          // pairs (slot of X_I,K, slot of X_J,K or -1: the chain workgroup's diagonal tile only)
    int32_t *back0 = nullptr, *back = nullptr;   // backward substitution: per column [back0[b], back0[b + 1]) pairs (block row i, slot of (i, b))
    int32_t *rhs_slot = nullptr;     // [nb] slot of the right-hand side's tile of a column
    // the exchange of several ranks: co-visible camera blocks (a, b <= a): block k = (cov_row[k], cov_adj[k]), camera a's at cov_start[a] ..
    int32_t *cov_start = nullptr, *cov_adj = nullptr, *cov_row = nullptr;
    double *W = nullptr;             // [n_tiles][64][64] the assembled tiles
    double *W2 = nullptr;            // [n_tiles][64][64] the factor's tiles X_IJ
    double *Ldiag = nullptr;         // [nb][64][64] inverses of the diagonal tiles' factors
    double *ybuf = nullptr;          // [nb * 64] solution in permuted order (a block is its own flag)
    int *xready = nullptr;           // [n_tiles] factor tile in memory
    int *ready = nullptr, *rpart = nullptr;   // [nb] inverse in memory / its rows 0..31 (count of 3)
};

struct SparseSolve;          // host object: plan + SparseDev + allocations (owned by esfm_ba_problem)
int ba_sparse_create(hipStream_t st, const SparsePlan &plan, const CamGraph &g, SparseSolve **out);
void ba_sparse_destroy(SparseSolve *S);
const SparsePlan &ba_sparse_plan_of(const SparseSolve *S);
// d.red (fixed point) -> assembled tiles -> factorisation -> backward substitution -> d.y_c.  After ba_sparse_pack + all-reduce the
// blocks are taken from the packed buffer instead.
int ba_solve_reduced_sparse(hipStream_t st, const BADev &d, SparseSolve *S, double radius, double min_diag, double max_diag);
// several ranks: this rank's co-visible blocks and right-hand side as doubles (36 per block, then 6 n_cam), d.red left all zeros
size_t ba_sparse_packed_doubles(const SparseSolve *S, int n_cam);
int ba_sparse_pack(hipStream_t st, const BADev &d, SparseSolve *S, double *packed);

}  // namespace esfm
