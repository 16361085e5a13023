# config-1 / config-3 end-to-end legs of bench.py alone (bin/sfm_native on the fountain images): wall seconds and the driver's stage split
import sys, json; sys.path.insert(0, '.')
import bench
for rep in range(2):
    for tag, f, p in (('config1', 'S', 300), ('config3', 'O', 8000)):
        r = bench.e2e_leg(tag, f, p, False)
        print(tag, 'wall', round(r['value'], 3), {k: round(v, 4) for k, v in r['stage_seconds'].items() if k in ('import+undistort', 'detect', 'match', 'verify', 'tracks', 'register', 'ba', 'sor', 'total')})
