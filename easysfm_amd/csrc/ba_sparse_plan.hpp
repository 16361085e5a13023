// Structure of the reduced camera system (host side, no GPU): which camera blocks of S = F'F + D^2 - sum_p (F'E) M^-1 (E'F) can be
// non-zero, an elimination order that keeps the Cholesky factor sparse AND its dependency chain short, and the resulting plan on
// 64 x 64 tiles for ba_chol_sparse.hip.
//
// The reference hands Ceres one residual block per observation (cpp_code/src/ba.cpp:140-151), so which cameras share a point is
// known before the solve; DENSE_SCHUR (ba.cpp:201) ignores it and factors the whole 6 Nc x 6 Nc matrix.  Block (a, b) of S is
// structurally non-zero only if cameras a and b observe a common point.  On a sequence capture (BASELINE config 5: every point seen
// by 10 consecutive cameras of a closed loop of 512) that is 4 % of the blocks -- and, ordered by nested dissection, the factor's
// longest dependency chain is 8 tile columns instead of 48.
#pragma once

#include <cstdint>
#include <vector>

namespace esfm {

// camera co-visibility: CSR adjacency (symmetric, without self loops), neighbours ascending
struct CamGraph {
    int n = 0;
    std::vector<int32_t> start, adj;
    bool has_edge(int a, int b) const;
};
// from observations sorted by point (CSR pt_start over obs_cam); `extra`: lower-triangle bit flags [a (a + 1) / 2 + b] to OR in
// (the other ranks' co-visibility), or NULL
CamGraph cam_graph_from_tracks(int n_cam, int n_pt, const int32_t *pt_start, const int32_t *obs_cam, const std::vector<uint8_t> *extra = nullptr);
// the lower-triangle flags of this rank's observations alone (what the ranks exchange)
std::vector<uint8_t> cam_pair_flags(int n_cam, int n_pt, const int32_t *pt_start, const int32_t *obs_cam);

struct SparsePlan {
    static constexpr int kTile = 64;
    int n_cam = 0;
    int nb = 0;                         // tile columns (the right-hand side is tile row nb)
    std::vector<int32_t> col_src;       // [nb * 64] original index 6 cam + a of a permuted column, -1: identity padding
    std::vector<int32_t> node_first_col, node_kind;   // supernodes in elimination order (first tile column, 0 leaf / 1 separator); node_first_col has one more entry
    // tiles of the factor (lower triangle incl. diagonal, symbolic fill included) + the right-hand side row; slot = index here
    struct Tile { int32_t I, J; };
    std::vector<Tile> tiles;
    std::vector<int32_t> slot_of;       // [(nb + 1) * nb] slot of tile (I, J), -1 outside the fill
    // workgroups of the factorisation in dispatch order (column by column).  A workgroup owns tile (I, J), subtracts
    // X_I,K X_J,K' for its operand list, then -- diagonal tile of a column nothing precedes: factors it -- or waits for
    // L_JJ^-1 and stores X_IJ; the LAST off-diagonal tile of block row I (a "chain" workgroup) goes on to finish and factor
    // diagonal tile (I, I): every X_I,K of its list is also subtracted (X X') from that tile, and the list holds ALL columns
    // K < J of block row I -- those block row J lacks with second slot -1 (no update of the own tile).
    struct Wg {
        int32_t I, J, slot;
        int32_t upd0, upd1;             // range in upd: pairs (slot of X_I,K, slot of X_J,K or -1), in expected order of availability
        int32_t dslot;                  // chain workgroups: slot of diagonal tile (I, I); -1 otherwise
        int32_t kind;                   // 0 off-diagonal, 1 stand-alone diagonal, 2 chain (off-diagonal + next diagonal), 3 right-hand-side row
    };
    std::vector<Wg> wgs;
    std::vector<int32_t> upd;
    // backward substitution: column b folds y_i for the rows i of its column structure (back0[b] .. back0[b + 1]): pairs (i, slot of (i, b))
    std::vector<int32_t> back0, back;
    int chain = 0;                      // longest dependency chain in tile columns
    long long update_steps = 0;         // 64 x 64 x 64 products of the factorisation
    int dense_nb = 0;                   // tile columns of the dense path, for comparison
    long long dense_tiles() const { return (long long)dense_nb * (dense_nb + 1) / 2 + dense_nb; }
    bool worthwhile() const;            // at most half the dense path's tiles, or at most half its dependency chain
};

// leaf_max: largest camera set that is not dissected further
SparsePlan make_sparse_plan(const CamGraph &g, int leaf_max = 32);

}  // namespace esfm
