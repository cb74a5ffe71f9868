"""Static check of one rule the compiler cannot apply inside inline asm: a DPP instruction must not read a VGPR that a VALU instruction wrote
fewer than two wait states earlier (s_nop N counts N + 1).  By default the rule is applied to the operand that goes through the DPP network
(src0) - the PGS turn's v_max_f32_dpp reads the residual (src1, an ordinary VGPR read with the ordinary interlock) right after the fmac that
wrote it; scripts/ubench/pgs2.hip variant 10 checks on the hardware that this gives the v_readlane form's numbers.  --strict applies it to every
VGPR a DPP instruction reads, as the compiler does for its own code.  usage: python scripts/dpp_hazards.py [--strict] file.s"""
import re
import sys


def _regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return [int(m.group(1))] if m else []


def check(path, strict=False):
    """(number of DPP instructions, list of violations as text) of an assembly listing."""
    lines = [l.strip() for l in open(path) if l.strip() and not l.strip().startswith(('.', ';', '//')) and not l.strip().endswith(':')]
    bad, n = [], 0
    for i, l in enumerate(lines):
        if '_dpp' not in l:
            continue
        n += 1
        op, rest = l.split(None, 1)
        toks = [t.strip() for t in re.split(r' (?:row_|quad_perm|wave_)', rest)[0].split(',')]
        if strict:                          # the compiler's own rule: every VGPR a DPP instruction reads
            srcs = [r for t in toks[1:] for r in _regs(t)]
            if op.startswith('v_fmac') or 'bank_mask:0xf' not in l or 'row_mask:0xf' not in l:
                srcs += _regs(toks[0])      # the accumulator / the lanes a mask keeps
        else:                               # the operand that goes through the DPP network (src0); the others are ordinary VGPR reads
            srcs = _regs(toks[1]) if len(toks) > 1 else []
        waits = 0
        for back in (1, 2):
            if i - back < 0:
                break
            p = lines[i - back]
            if p.startswith('v_'):
                pd = _regs(p.split(None, 1)[1].split(',')[0].strip())
                if waits < 2 and any(r in pd for r in srcs):
                    bad.append(f'wait states {waits}: {p}   ->   {l}')
            m = re.match(r's_nop (\d+)', p)
            waits += int(m.group(1)) + 1 if m else 1
    return n, bad


if __name__ == '__main__':
    n, bad = check(sys.argv[-1], strict='--strict' in sys.argv)
    for b in bad[:20]:
        print(b)
    print('dpp instructions', n, 'violations', len(bad))
