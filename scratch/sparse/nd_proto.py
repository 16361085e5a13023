"""Prototype of the structure-aware reduced solve's host planning: camera co-visibility graph -> automatic nested
dissection (BFS level sets) -> supernodes padded to 64-column tiles -> symbolic tile Cholesky -> tile count, flops, critical chain."""
import sys, numpy as np
sys.path.insert(0, '.')
from easysfm_amd import synth

CB = 64

def covis(n_cam, cam_idx, pt_idx, n_pt):
    order = np.argsort(pt_idx, kind='stable')
    c = cam_idx[order]; p = pt_idx[order]
    start = np.searchsorted(p, np.arange(n_pt + 1))
    A = np.zeros((n_cam, n_cam), bool)
    lens = np.diff(start)
    # vectorised for equal-length tracks
    for T in np.unique(lens):
        if T == 0: continue
        sel = np.nonzero(lens == T)[0]
        cams = c[(start[sel][:, None] + np.arange(T)[None, :])]
        for a in range(T):
            for b in range(T):
                A[cams[:, a], cams[:, b]] = True
    return A

def bfs_levels(adj, nodes_mask, root):
    lev = -np.ones(len(adj), int); lev[root] = 0
    cur = [root]; levels = [cur]
    while True:
        nxt = []
        for u in cur:
            for v in adj[u]:
                if nodes_mask[v] and lev[v] < 0:
                    lev[v] = len(levels); nxt.append(v)
        if not nxt: break
        levels.append(nxt); cur = nxt
    return levels

def pseudo_peripheral(adj, mask, start):
    r = start; levels = bfs_levels(adj, mask, r)
    while True:
        last = levels[-1]
        cand = min(last, key=lambda v: sum(1 for w in adj[v] if mask[w]))
        l2 = bfs_levels(adj, mask, cand)
        if len(l2) > len(levels): r, levels = cand, l2
        else: return r, levels

def components(adj, mask):
    seen = np.zeros(len(adj), bool); out = []
    for s in np.nonzero(mask)[0]:
        if seen[s]: continue
        comp = [s]; seen[s] = True; k = 0
        while k < len(comp):
            u = comp[k]; k += 1
            for v in adj[u]:
                if mask[v] and not seen[v]: seen[v] = True; comp.append(v)
        out.append(comp)
    return out

def nested_dissection(adj, n, leaf_max):
    nodes = []   # supernodes in elimination order: (list of cameras, kind)
    def dissect(comp):
        mask = np.zeros(n, bool); mask[comp] = True
        r, levels = pseudo_peripheral(adj, mask, comp[0])
        flat = [v for L in levels for v in L]
        if len(comp) <= leaf_max or len(levels) < 3:
            nodes.append((flat, 'leaf')); return
        sizes = np.array([len(L) for L in levels]); cum = np.cumsum(sizes)
        # middle level: minimise |left - right|, interior levels only
        best = min(range(1, len(levels) - 1), key=lambda m: (abs((cum[m - 1]) - (cum[-1] - cum[m])), sizes[m]))
        S = levels[best]
        if len(S) * 3 > len(comp):
            nodes.append((flat, 'leaf')); return
        mask[S] = False
        for cc in components(adj, mask):
            dissect(cc)
        nodes.append((list(S), 'sep'))
    mask = np.ones(n, bool)
    for cc in components(adj, mask):
        dissect(cc)
    return nodes

def plan(A, leaf_max=32):
    n = len(A)
    adj = [list(np.nonzero(A[i])[0][np.nonzero(A[i])[0] != i]) for i in range(n)]
    nodes = nested_dissection(adj, n, leaf_max)
    # columns: node by node, padded to tiles
    col_cam = []
    tile_node = []
    for k, (cams, kind) in enumerate(nodes):
        cols = [(c, a) for c in cams for a in range(6)]
        pad = (-len(cols)) % CB
        cols += [(-1, 0)] * pad
        col_cam += cols
        tile_node += [k] * (len(cols) // CB)
    nb = len(col_cam) // CB
    camt = [[] for _ in range(nb)]
    for i, (c, a) in enumerate(col_cam):
        if c >= 0 and (not camt[i // CB] or camt[i // CB][-1] != c): camt[i // CB].append(c)
    L = np.zeros((nb, nb), bool)
    for I in range(nb):
        for J in range(I + 1):
            L[I, J] = I == J or A[np.ix_(camt[I], camt[J])].any()
    orig = L.sum()
    for J in range(nb):
        rows = np.nonzero(L[J + 1:, J])[0] + J + 1
        for a in rows:
            L[a, rows[rows <= a]] = True
    # critical path (in tile steps): finish[J] = 1 + max over K in rowstruct(J) finish[K]
    fin = np.zeros(nb, int)
    for J in range(nb):
        ks = np.nonzero(L[J, :J])[0]
        fin[J] = 1 + (fin[ks].max() if len(ks) else 0)
    # flops: update steps (2*64^3 each), trsm products (64^3), factor
    upd = 0
    for J in range(nb):
        for I in np.nonzero(L[J:, J])[0] + J:
            upd += int((L[I, :J] & L[J, :J]).sum())
    ntiles = int(L.sum())
    return dict(nodes=[(len(c), k) for c, k in nodes], nb=nb, n_pad=nb * CB, tiles=ntiles, tiles_orig=int(orig), dense_tiles=(6 * n + CB - 1) // CB * ((6 * n + CB - 1) // CB + 1) // 2,
                update_steps=upd, chain=int(fin.max()), fin=fin, L=L)

if __name__ == '__main__':
    for name, (nc, npt, k, rad, ext, seed) in {'BA-512': (512, 300000, 10, 40.0, 8.0, 5000), 'BA-107': (107, 3000, 6, 10, 2, 1), 'BA-43': (43, 2000, 5, 10, 2, 1)}.items():
        sc = synth.ba_scene(nc, npt, k, radius=rad, extent=ext, seed=seed)
        A = covis(nc, sc.cam_idx, sc.pt_idx, npt)
        occ = A.sum() / A.size
        for leaf_max in (16, 32, 64):
            p = plan(A, leaf_max)
            print(f"{name}: camera-block occupancy {occ:.3f}; leaf_max {leaf_max}: {len(p['nodes'])} supernodes, nb {p['nb']} (n_pad {p['n_pad']} vs {6*nc}), tiles in fill {p['tiles']} "
                  f"(orig {p['tiles_orig']}, dense {p['dense_tiles']}), update steps {p['update_steps']} (dense {sum((nbd - j) * j for nbd in [ (6*nc+63)//64 ] for j in range(nbd)) }), chain {p['chain']} tile steps")
        print('   nodes', p['nodes'])
