"""Static instruction count of a kernel's hottest loop (the backward-branch span that holds most global stores): python scratch/count_loop.py file.s mangled-name-substring"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
name = [m for m in re.findall(r'^(_Z\w+):', s, re.M) if sys.argv[2] in m][0]
a = s.index(name + ':'); b = s.index('.Lfunc_end', a)
lines = s[a:b].splitlines()
pos, ins = {}, []
for l in lines:
    m = re.match(r'^(\.LBB\w+):', l)
    if m: pos[m.group(1)] = len(ins); continue
    t = l.strip()
    if l.startswith('\t') and t and not t.startswith('.') and not t.startswith(';'): ins.append(t)
best = None
for i, t in enumerate(ins):
    m = re.match(r's_cbranch\w*\s+(\.LBB\w+)', t) or re.match(r's_branch\s+(\.LBB\w+)', t)
    if m and m.group(1) in pos and pos[m.group(1)] <= i:
        span = ins[pos[m.group(1)]:i + 1]
        st = sum(1 for x in span if x.startswith('global_store'))
        if best is None or st > best[0] or (st == best[0] and len(span) < len(best[1])): best = (st, span)
print(name, 'kernel', len(ins), 'loop', len(best[1]), 'stores', best[0])
c = Counter(x.split()[0] for x in best[1])
print(' '.join(f'{k}:{v}' for k, v in c.most_common(30)))
