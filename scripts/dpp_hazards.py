"""Static check of one rule the compiler cannot apply inside inline asm: a DPP instruction must not read a VGPR that a VALU instruction wrote
fewer than two wait states earlier (s_nop N counts N + 1).  usage: python scripts/dpp_hazards.py file.s   (hipcc -S --cuda-device-only output)"""
import re
import sys
L = [l.strip() for l in open(sys.argv[1]) if l.strip() and not l.strip().startswith(('.', ';', '//')) and not l.strip().endswith(':')]


def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return [int(m.group(1))] if m else []


bad = n = 0
for i, l in enumerate(L):
    if '_dpp' not in l:
        continue
    n += 1
    op, rest = l.split(None, 1)
    toks = [t.strip() for t in re.split(r' (?:row_|quad_perm|wave_)', rest)[0].split(',')]
    srcs = [r for t in toks[1:] for r in regs(t)]
    if op.startswith('v_fmac') or 'bank_mask:0xf' not in l or 'row_mask:0xf' not in l:
        srcs += regs(toks[0])          # the accumulator / the lanes a mask keeps
    waits = 0
    for back in (1, 2):
        if i - back < 0:
            break
        p = L[i - back]
        if p.startswith('v_'):
            pd = regs(p.split(None, 1)[1].split(',')[0].strip())
            if waits < 2 and any(r in pd for r in srcs):
                bad += 1
                if bad <= 20:
                    print(f'wait states {waits}: {p}   ->   {l}')
        m = re.match(r's_nop (\d+)', p)
        waits += int(m.group(1)) + 1 if m else 1
print('dpp instructions', n, 'violations', bad)
