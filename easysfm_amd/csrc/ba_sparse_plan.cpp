// Host-side planning of the structure-aware reduced camera solve (see ba_sparse_plan.hpp).  No GPU code here.
#include "ba_sparse_plan.hpp"

#include <algorithm>
#include <cstddef>

namespace esfm {

bool CamGraph::has_edge(int a, int b) const
{
    if (a == b) return true;
    const int32_t *lo = adj.data() + start[(size_t)a], *hi = adj.data() + start[(size_t)a + 1];
    return std::binary_search(lo, hi, (int32_t)b);
}

std::vector<uint8_t> cam_pair_flags(int n_cam, int n_pt, const int32_t *pt_start, const int32_t *obs_cam)
{
    std::vector<uint8_t> f((size_t)n_cam * ((size_t)n_cam + 1) / 2, 0);
    std::vector<int32_t> cams;
    for (int p = 0; p < n_pt; ++p) {
        const int b = pt_start[p], e = pt_start[p + 1];
        if (e - b < 2) continue;
        cams.assign(obs_cam + b, obs_cam + e);
        std::sort(cams.begin(), cams.end());
        cams.erase(std::unique(cams.begin(), cams.end()), cams.end());
        for (size_t i = 1; i < cams.size(); ++i) {
            const size_t row = (size_t)cams[i] * ((size_t)cams[i] + 1) / 2;
            for (size_t j = 0; j < i; ++j) f[row + (size_t)cams[j]] = 1;
        }
    }
    return f;
}

CamGraph cam_graph_from_tracks(int n_cam, int n_pt, const int32_t *pt_start, const int32_t *obs_cam, const std::vector<uint8_t> *extra)
{
    std::vector<uint8_t> f = cam_pair_flags(n_cam, n_pt, pt_start, obs_cam);
    if (extra && extra->size() == f.size())
        for (size_t k = 0; k < f.size(); ++k) f[k] |= (*extra)[k];
    CamGraph g;
    g.n = n_cam;
    std::vector<int32_t> deg((size_t)n_cam, 0);
    for (int a = 1; a < n_cam; ++a) {
        const size_t row = (size_t)a * ((size_t)a + 1) / 2;
        for (int b = 0; b < a; ++b) if (f[row + (size_t)b]) { ++deg[(size_t)a]; ++deg[(size_t)b]; }
    }
    g.start.assign((size_t)n_cam + 1, 0);
    for (int a = 0; a < n_cam; ++a) g.start[(size_t)a + 1] = g.start[(size_t)a] + deg[(size_t)a];
    g.adj.resize((size_t)g.start[(size_t)n_cam]);
    std::vector<int32_t> fill(g.start.begin(), g.start.end() - 1);
    // ascending neighbour lists: first the lower neighbours of a (b < a, ascending), then a as lower neighbour of the later rows
    for (int a = 0; a < n_cam; ++a) {
        const size_t row = (size_t)a * ((size_t)a + 1) / 2;
        for (int b = 0; b < a; ++b) if (f[row + (size_t)b]) g.adj[(size_t)fill[(size_t)a]++] = b;
    }
    for (int a = 0; a < n_cam; ++a) {
        const size_t row = (size_t)a * ((size_t)a + 1) / 2;
        for (int b = 0; b < a; ++b) if (f[row + (size_t)b]) g.adj[(size_t)fill[(size_t)b]++] = a;
    }
    return g;
}

namespace {

// Automatic nested dissection on breadth-first level structures (George & Liu): a component is split by the middle level of the
// level structure rooted at a pseudo-peripheral vertex; the parts are numbered first (recursively), the separator last.  A
// component of at most leaf_max cameras -- or one whose level structure has no usable middle (fewer than three levels, or a
// separator larger than a third of it: a nearly complete graph) -- becomes a leaf, numbered in level order (a banded leaf).
struct Dissector {
    const CamGraph &g;
    int leaf_max;
    std::vector<int32_t> comp_of;              // current component id of every vertex (-1: already numbered as part of a separator)
    int next_comp = 1;
    std::vector<std::vector<int32_t>> nodes;   // supernodes in elimination order
    std::vector<int32_t> kinds;
    std::vector<int32_t> level;                // scratch

    // level structure of component `c` rooted at r: vertices in level order, level_start offsets
    void levels_from(int r, int c, std::vector<int32_t> &order, std::vector<int32_t> &lstart)
    {
        order.clear(); lstart.clear();
        order.push_back(r); level[(size_t)r] = 0; lstart.push_back(0);
        size_t head = 0;
        int cur = 0;
        while (head < order.size()) {
            const int u = order[head];
            if (level[(size_t)u] != cur) { cur = level[(size_t)u]; lstart.push_back((int32_t)head); }
            ++head;
            for (int32_t k = g.start[(size_t)u]; k < g.start[(size_t)u + 1]; ++k) {
                const int v = g.adj[(size_t)k];
                if (comp_of[(size_t)v] == c && level[(size_t)v] < 0) { level[(size_t)v] = level[(size_t)u] + 1; order.push_back(v); }
            }
        }
        lstart.push_back((int32_t)order.size());
        for (int v : order) level[(size_t)v] = -1;
    }
    int degree_in(int v, int c) const
    {
        int dg = 0;
        for (int32_t k = g.start[(size_t)v]; k < g.start[(size_t)v + 1]; ++k) dg += comp_of[(size_t)g.adj[(size_t)k]] == c;
        return dg;
    }
    void dissect(const std::vector<int32_t> &verts, int c)
    {
        std::vector<int32_t> order, lstart, order2, lstart2;
        int r = verts[0];
        levels_from(r, c, order, lstart);
        for (int guard = 0; guard < 16; ++guard) {          // pseudo-peripheral root: restart from a minimum-degree vertex of the last level while the structure deepens
            int best = -1, best_deg = 0;
            for (int32_t k = lstart[lstart.size() - 2]; k < lstart.back(); ++k) {
                const int dg = degree_in(order[(size_t)k], c);
                if (best < 0 || dg < best_deg) { best = order[(size_t)k]; best_deg = dg; }
            }
            levels_from(best, c, order2, lstart2);
            if (lstart2.size() > lstart.size()) { order.swap(order2); lstart.swap(lstart2); r = best; } else break;
        }
        const int n_levels = (int)lstart.size() - 1;
        const int nv = (int)order.size();
        int split = -1;
        if (nv > leaf_max && n_levels >= 3) {
            long best_cost = -1;
            for (int m = 1; m + 1 < n_levels; ++m) {
                const long left = lstart[(size_t)m], right = nv - lstart[(size_t)m + 1], sz = lstart[(size_t)m + 1] - lstart[(size_t)m];
                const long cost = (left > right ? left - right : right - left) * 4096 + sz;
                if (best_cost < 0 || cost < best_cost) { best_cost = cost; split = m; }
            }
            if (3 * (lstart[(size_t)split + 1] - lstart[(size_t)split]) > nv) split = -1;
        }
        if (split < 0) { nodes.push_back(order); kinds.push_back(0); for (int v : order) comp_of[(size_t)v] = -1; return; }
        std::vector<int32_t> sep(order.begin() + lstart[(size_t)split], order.begin() + lstart[(size_t)split + 1]);
        for (int v : sep) comp_of[(size_t)v] = -1;
        // the remaining vertices fall into components (at least two: the levels before and after the separator)
        std::vector<int32_t> rest;
        for (int v : order) if (comp_of[(size_t)v] == c) rest.push_back(v);
        std::vector<int32_t> comp, stack;
        for (int s : rest) {
            if (comp_of[(size_t)s] != c) continue;
            const int nc = next_comp++;
            comp.clear(); stack.clear();
            comp_of[(size_t)s] = nc; stack.push_back(s);
            while (!stack.empty()) {
                const int u = stack.back(); stack.pop_back();
                comp.push_back(u);
                for (int32_t k = g.start[(size_t)u]; k < g.start[(size_t)u + 1]; ++k) {
                    const int v = g.adj[(size_t)k];
                    if (comp_of[(size_t)v] == c) { comp_of[(size_t)v] = nc; stack.push_back(v); }
                }
            }
            std::sort(comp.begin(), comp.end());
            const std::vector<int32_t> mine = comp;       // (comp is reused by the siblings)
            dissect(mine, nc);
        }
        nodes.push_back(sep); kinds.push_back(1);
    }
};

}  // namespace

// The dense path stays when the structure saves less than half of its tiles AND less than half of its dependency chain (BA-25, the
// fountain: every camera sees every other one).  The chain is what the dense factorisation is bound by (one tile column every
// ~15 us), so halving it pays even where the tile count does not halve (200 cameras of a loop: chain 7 against 19, 108 of 209 tiles).
bool SparsePlan::worthwhile() const
{
    return nb > 0 && (2 * (long long)tiles.size() <= dense_tiles() || (2 * chain <= dense_nb && (long long)tiles.size() <= 2 * dense_tiles()));
}

SparsePlan make_sparse_plan(const CamGraph &g, int leaf_max)
{
    SparsePlan P;
    constexpr int T = SparsePlan::kTile;
    P.n_cam = g.n;
    P.dense_nb = (6 * g.n + T - 1) / T;
    if (g.n <= 0) return P;
    Dissector D{g, std::max(leaf_max, 1), std::vector<int32_t>((size_t)g.n, 0), 1, {}, {}, std::vector<int32_t>((size_t)g.n, -1)};
    // connected components of the whole graph first (id 0 = "not yet assigned").  Components of at most leaf_max cameras -- cameras
    // without observations, small groups that share nothing with the rest -- are packed TOGETHER into leaves of at most leaf_max
    // cameras: they do not touch each other, so sharing tiles costs nothing, and a tile of padding per isolated camera is saved.
    {
        std::vector<int32_t> comp, stack, small;
        auto flush_small = [&]() {
            if (small.empty()) return;
            D.nodes.push_back(small); D.kinds.push_back(0);
            for (int v : small) D.comp_of[(size_t)v] = -1;
            small.clear();
        };
        for (int s = 0; s < g.n; ++s) {
            if (D.comp_of[(size_t)s] != 0) continue;
            const int nc = D.next_comp++;
            comp.clear(); stack.clear();
            D.comp_of[(size_t)s] = nc; stack.push_back(s);
            while (!stack.empty()) {
                const int u = stack.back(); stack.pop_back();
                comp.push_back(u);
                for (int32_t k = g.start[(size_t)u]; k < g.start[(size_t)u + 1]; ++k) {
                    const int v = g.adj[(size_t)k];
                    if (D.comp_of[(size_t)v] == 0) { D.comp_of[(size_t)v] = nc; stack.push_back(v); }
                }
            }
            std::sort(comp.begin(), comp.end());
            if ((int)comp.size() <= D.leaf_max) {
                if ((int)(small.size() + comp.size()) > D.leaf_max) flush_small();
                small.insert(small.end(), comp.begin(), comp.end());
                continue;
            }
            const std::vector<int32_t> mine = comp;
            D.dissect(mine, nc);
        }
        flush_small();
    }
    // columns: supernode by supernode, each padded to whole tiles
    std::vector<int32_t> cam_tile0((size_t)g.n, 0), cam_tile1((size_t)g.n, 0);      // first / last tile holding columns of the camera
    for (size_t k = 0; k < D.nodes.size(); ++k) {
        P.node_first_col.push_back((int32_t)(P.col_src.size() / T));
        P.node_kind.push_back(D.kinds[k]);
        for (int c : D.nodes[k]) {
            cam_tile0[(size_t)c] = (int32_t)(P.col_src.size() / T);
            for (int a = 0; a < 6; ++a) P.col_src.push_back(6 * c + a);
            cam_tile1[(size_t)c] = (int32_t)((P.col_src.size() - 1) / T);
        }
        while (P.col_src.size() % T) P.col_src.push_back(-1);
    }
    P.nb = (int)(P.col_src.size() / T);
    P.node_first_col.push_back(P.nb);
    const int nb = P.nb;
    // tile structure of S, then symbolic fill (right-looking: eliminating column J joins the rows of its structure pairwise)
    std::vector<uint8_t> L((size_t)nb * nb, 0);
    auto mark = [&](int i, int j) { if (i < j) std::swap(i, j); L[(size_t)i * nb + j] = 1; };
    for (int a = 0; a < g.n; ++a) {
        for (int ta = cam_tile0[(size_t)a]; ta <= cam_tile1[(size_t)a]; ++ta) {
            for (int tb = cam_tile0[(size_t)a]; tb <= cam_tile1[(size_t)a]; ++tb) mark(ta, tb);
            for (int32_t k = g.start[(size_t)a]; k < g.start[(size_t)a + 1]; ++k) {
                const int b = g.adj[(size_t)k];
                for (int tb = cam_tile0[(size_t)b]; tb <= cam_tile1[(size_t)b]; ++tb) mark(ta, tb);
            }
        }
    }
    for (int j = 0; j < nb; ++j) L[(size_t)j * nb + j] = 1;       // (tiles of padding only)
    std::vector<int32_t> rows;
    for (int j = 0; j < nb; ++j) {
        rows.clear();
        for (int i = j + 1; i < nb; ++i) if (L[(size_t)i * nb + j]) rows.push_back(i);
        for (size_t x = 0; x < rows.size(); ++x)
            for (size_t y = 0; y <= x; ++y) L[(size_t)rows[x] * nb + rows[y]] = 1;
    }
    // slots: column by column, diagonal first, the right-hand side row last
    P.slot_of.assign((size_t)(nb + 1) * nb, -1);
    for (int j = 0; j < nb; ++j) {
        for (int i = j; i < nb; ++i)
            if (L[(size_t)i * nb + j]) { P.slot_of[(size_t)i * nb + j] = (int32_t)P.tiles.size(); P.tiles.push_back({i, j}); }
        P.slot_of[(size_t)nb * nb + j] = (int32_t)P.tiles.size(); P.tiles.push_back({nb, j});
    }
    auto slot = [&](int i, int j) { return P.slot_of[(size_t)i * nb + j]; };
    // last column of every block row's structure left of the diagonal: that tile's workgroup finishes the diagonal tile
    std::vector<int32_t> klast((size_t)nb, -1), fin((size_t)nb, 0);
    for (int i = 0; i < nb; ++i)
        for (int k = 0; k < i; ++k) if (L[(size_t)i * nb + k]) klast[(size_t)i] = k;
    for (int j = 0; j < nb; ++j) {
        int f = 0;
        for (int k = 0; k < j; ++k) if (L[(size_t)j * nb + k]) f = std::max(f, fin[(size_t)k]);
        fin[(size_t)j] = f + 1;
        P.chain = std::max(P.chain, f + 1);
    }
    // A workgroup's operand tiles are listed in the order they are expected to BECOME AVAILABLE -- by the depth fin[K] of their
    // column in the dependency chain, then by column -- not by column alone: subtrees of the elimination tree run side by side, and
    // a separator's workgroup that waited for an early separator's tile with a dozen long-finished leaf tiles queued behind it
    // did all of them AFTER the tile it was really waiting for (BA-512: 110 of the kernel's 217 us).  The order is part of the plan,
    // so the sums keep a fixed order.
    std::vector<int32_t> ks;
    auto by_availability = [&](std::vector<int32_t> &v) {
        std::stable_sort(v.begin(), v.end(), [&](int32_t a, int32_t b) { return fin[(size_t)a] != fin[(size_t)b] ? fin[(size_t)a] < fin[(size_t)b] : a < b; });
    };
    for (int j = 0; j < nb; ++j) {
        if (klast[(size_t)j] < 0) P.wgs.push_back({j, j, slot(j, j), (int32_t)P.upd.size(), (int32_t)P.upd.size(), -1, 1});
        for (int i = j + 1; i <= nb; ++i) {
            if (i < nb && !L[(size_t)i * nb + j]) continue;
            SparsePlan::Wg w;
            w.I = i; w.J = j; w.slot = slot(i, j);
            w.dslot = -1;
            w.kind = i == nb ? 3 : 0;
            const bool chain = i < nb && klast[(size_t)i] == j;
            // operand columns K < J: those both block rows have (the tile's own updates); a chain workgroup also takes every other
            // column of block row I (updates of diagonal tile (I, I) only: second slot -1)
            ks.clear();
            for (int k = 0; k < j; ++k)
                if ((i == nb || L[(size_t)i * nb + k]) && (L[(size_t)j * nb + k] || chain)) ks.push_back(k);
            by_availability(ks);
            w.upd0 = (int32_t)P.upd.size();
            for (int k : ks) {
                const bool own = L[(size_t)j * nb + k] != 0;
                P.upd.push_back(slot(i, k)); P.upd.push_back(own ? slot(j, k) : -1);
                P.update_steps += (own ? 1 : 0) + (chain ? 1 : 0);
            }
            w.upd1 = (int32_t)P.upd.size();
            if (chain) { w.kind = 2; w.dslot = slot(i, i); ++P.update_steps; }
            P.wgs.push_back(w);
        }
    }
    P.back0.assign((size_t)nb + 1, 0);
    for (int b = 0; b < nb; ++b) {
        P.back0[(size_t)b] = (int32_t)(P.back.size() / 2);
        for (int i = nb - 1; i > b; --i) if (L[(size_t)i * nb + b]) { P.back.push_back(i); P.back.push_back(slot(i, b)); }
    }
    P.back0[(size_t)nb] = (int32_t)(P.back.size() / 2);
    return P;
}

}  // namespace esfm
