import os, subprocess, sys, numpy as np
from PIL import Image
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = '/tmp/cmpdrv'; os.makedirs(tmp + '/images', exist_ok=True)
z = np.load(os.path.join(root, 'tests/golden/fountain11_half_gray.npz'))
names = []
for i, img in enumerate(z['images'][:6]):
    names.append(f'{i:04d}.png'); Image.fromarray(np.stack([img, np.roll(img, 1, 1), img // 2 + 60], axis=2)).save(f'{tmp}/images/{names[-1]}')
open(tmp + '/list.txt', 'w').write('\n'.join(names) + '\n')
open(tmp + '/K.txt', 'w').write(f'{689.87 / 2} 0 {380.17 / 2}\n0 {691.04 / 2} {251.70 / 2}\n0 0 1\n')
args = [tmp + '/images', tmp + '/list.txt', tmp + '/K.txt', 'none']; tail = ['S', '100', '1.0', '1', '0', '4', '1', '0']
for name, cmd in (('c', [os.path.join(root, 'bin/sfm_native')]), ('p', [sys.executable, os.path.join(root, 'bin/sfm')]), ('p2', [sys.executable, os.path.join(root, 'bin/sfm')])):
    r = subprocess.run(cmd + args + [f'{tmp}/{name}.ply'] + tail, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    keep = [l for l in r.stdout.splitlines() if any(k in l for k in ('inlier', 'Inlier', 'Triangulate', 'correspond', 'Progress', 'ilter', 'BA', 'Bundle', 'iteration', 'Initialization', 'PnP', 'reproj', 'Found', 'unique', 'cost', 'Output [', 'depth'))]
    open(f'{tmp}/{name}.log', 'w').write('\n'.join(keep))
    print(name, r.returncode, len(keep))
for n in ('c','p'):
    print('=====', n)
    print('\n'.join(l for l in open(f'{tmp}/{n}.log').read().splitlines() if 'match SURF' not in l and 'Find [' not in l and 'Found' not in l))
import difflib
a = open(tmp + '/c.log').read().splitlines(); b = open(tmp + '/p.log').read().splitlines(); c = open(tmp + '/p2.log').read().splitlines()
print('--- C++ vs Python'); print('\n'.join(list(difflib.unified_diff(a, b, lineterm='', n=0))[:60]))
print('--- Python vs Python'); print('\n'.join(list(difflib.unified_diff(b, c, lineterm='', n=0))[:20]))
