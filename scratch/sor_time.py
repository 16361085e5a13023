# SOR filter: sorted sweep window against the all-candidates sweep (ESFM_SOR_BRUTE=1 in a child process), bitwise comparison and timing
import os, sys, time, subprocess; sys.path.insert(0, '/root/repo')
import numpy as np
def run(tag):
    import easysfm_amd as E
    from easysfm_amd import _lib
    ctx = E.Context(0)
    out = {}
    rng = np.random.default_rng(7)
    for name, n in (('uniform30k', 30600), ('uniform200k', 200000), ('clustered100k', 100000), ('plane50k', 50000), ('line20k', 20000)):
        if name.startswith('uniform'): P = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
        elif name.startswith('clustered'):
            c = rng.uniform(-20, 20, (40, 3)); P = (c[rng.integers(0, 40, n)] + rng.normal(0, 0.3, (n, 3)) * rng.uniform(0.2, 3, (n, 1))).astype(np.float32)
            P[:2000] = rng.uniform(-60, 60, (2000, 3)).astype(np.float32)
        elif name.startswith('plane'): P = rng.uniform(-5, 5, (n, 3)).astype(np.float32); P[:, 0] = 1.25
        else: P = np.zeros((n, 3), np.float32); P[:, 1] = rng.uniform(-5, 5, n)
        P[17] = [np.nan, 0, 0]; P[123, 2] = np.inf
        E.sor_filter(P, 50, 2.0, ctx)
        ctx.set_kernel_timing(True); ctx.kernel_time(_lib.K_SOR_KNN)
        t = time.perf_counter(); keep, md, thr = E.sor_filter(P, 50, 2.0, ctx); el = time.perf_counter() - t
        ms, cnt = ctx.kernel_time(_lib.K_SOR_KNN); ctx.set_kernel_timing(False)
        out[name] = (md, thr, keep, ms / max(cnt, 1), el * 1e3)
    np.savez('/tmp/sor_%s.npz' % tag, **{k + '.md': v[0] for k, v in out.items()}, **{k + '.keep': v[2] for k, v in out.items()})
    for k, v in out.items(): print(f'{tag:6s} {k:14s} kernel {v[3]:8.3f} ms  call {v[4]:8.2f} ms  thr {v[1]:.9g}  kept {int(v[2].sum())}')
if len(sys.argv) > 1: run(sys.argv[1]); sys.exit(0)
subprocess.run([sys.executable, __file__, 'sorted'], check=True)
subprocess.run([sys.executable, __file__, 'brute'], check=True, env=dict(os.environ, ESFM_SOR_BRUTE='1'))
a, b = np.load('/tmp/sor_sorted.npz'), np.load('/tmp/sor_brute.npz')
for k in a.files:
    same = np.array_equal(a[k].view(np.uint32) if a[k].dtype == np.float32 else a[k], b[k].view(np.uint32) if b[k].dtype == np.float32 else b[k])
    print(k, 'bit-identical' if same else 'DIFFERENT: %d entries' % int((a[k] != b[k]).sum()))
