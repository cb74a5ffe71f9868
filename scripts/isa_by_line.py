"""Static instruction counts of one kernel by source line (the .loc directives of a -gline-tables-only assembly listing).
usage: python scripts/isa_by_line.py <listing.s> <mangled kernel name> [file index of interest = 1] [first line of the region]
Instructions inlined from helpers are attributed to the helper's own line; VALU = v_* opcodes."""
import collections
import re
import sys
path, kern = sys.argv[1], sys.argv[2]
fsel = int(sys.argv[3]) if len(sys.argv) > 3 else None
inside, cur = False, (0, 0)
cnt = collections.Counter(); valu = collections.Counter()
for ln in open(path):
    if ln.startswith(kern + ':'):
        inside = True
        continue
    if not inside:
        continue
    s = ln.strip()
    if s.startswith('.loc'):
        p = s.split()
        cur = (int(p[1]), int(p[2]))
        continue
    if s.startswith('s_endpgm'):
        break
    if not s or s[0] in '.;' or s.endswith(':'):
        continue
    op = s.split()[0]
    cnt[cur] += 1
    if op.startswith('v_'):
        valu[cur] += 1
tot = sum(valu.values())
print('total instructions', sum(cnt.values()), 'VALU', tot)
rows = sorted(valu.items(), key=lambda kv: -kv[1])
for (f, l), v in rows[:int(sys.argv[4]) if len(sys.argv) > 4 else 60]:
    if fsel is None or f == fsel or True:
        print(f'file {f} line {l:5d}: VALU {v:5d}  all {cnt[(f, l)]:5d}')
