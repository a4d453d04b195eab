#!/usr/bin/env python3
"""Static instruction histogram of the tiled kernel by source region (from `hipcc -gline-tables-only -S`).
usage: static_hist.py L31g.s <kernel-substring> [--lines]
Instructions of inlined helpers (lines < 172 of s2k_tile_impl.h, or other files) count for the last kernel-side line seen."""
import re, sys, collections
path, kern = sys.argv[1], sys.argv[2]
per_line = '--lines' in sys.argv
regions = [(172, 335, 'hash'), (366, 540, 'compact'), (541, 611, 'rawpos'), (612, 674, 'hpcsimd.lookback'), (675, 850, 'dense.boundaries'),
           (851, 900, 'dense.count'), (901, 945, 'dense.segments+word'), (946, 1005, 'dense.rederive'), (1006, 1030, 'dense.list'),
           (1031, 1060, 'dense.jobs'), (1061, 1158, 'dense.rounds'), (1159, 1500, 'kernel')]
def region(l):
    for a, b, n in regions:
        if a <= l <= b: return n
    return None
cur, curline, infn = 'kernel', 0, False
hist = collections.defaultdict(lambda: collections.Counter())
lines = collections.defaultdict(lambda: collections.Counter())
tile_file = None
for ln in open(path):
    s = ln.strip()
    m = re.match(r'\.file\s+(\d+)\s+"[^"]*"\s+"s2k_tile_impl\.h"', s)
    if m: tile_file = m.group(1)
    if re.match(r'_Z\S+:', s) and kern in s: infn = True; continue
    if infn and s.startswith('.Lfunc_end'): infn = False
    if not infn: continue
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
    if m:
        if m.group(1) == tile_file:
            r = region(int(m.group(2)))
            if r: cur, curline = r, int(m.group(2))
        continue
    if not s or s[0] in '.;' or re.match(r'\S+:', s): continue
    op = s.split()[0]
    kind = 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'vmem' if op.split('_')[0] in ('global', 'scratch', 'buffer', 'flat') else 'other'
    hist[cur][kind] += 1
    lines[(cur, curline)][kind] += 1
tot = collections.Counter()
for r, c in hist.items():
    print(f'{r:18s} ' + ' '.join(f'{k}={c[k]:5d}' for k in ('valu', 'salu', 'lds', 'vmem', 'other')))
    tot.update(c)
print('total', dict(tot))
if per_line:
    for (r, l), c in sorted(lines.items(), key=lambda x: x[0][1]):
        if c['valu'] + c['lds'] + c['salu'] >= 8: print(f'  {r:18s} line {l:5d}  valu={c["valu"]:4d} salu={c["salu"]:4d} lds={c["lds"]:4d} vmem={c["vmem"]:3d}')
