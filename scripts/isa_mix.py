"""Static instruction mix of a kernel between its s_barriers, from hipcc -S output.
Usage: python scripts/isa_mix.py file.s 'qp_reg_kernelILb0'   (substring of the mangled kernel name)"""
import re, sys
L = open(sys.argv[1]).read().split('\n'); key = sys.argv[2]
s = next(i for i, l in enumerate(L) if l.startswith('_Z') and key in l and l.rstrip().endswith(':') or (l.startswith('_Z') and key in l and ':' in l))
e = next(i for i in range(s, len(L)) if L[i].startswith('.Lfunc_end'))
seg = []; cur = []; start = s
for i in range(s, e):
    ln = L[i].strip()
    if not ln or ln.startswith(';') or (ln.startswith('.') and not ln.startswith('.LBB')): continue
    cur.append(ln)
    if ln.startswith('s_barrier'): seg.append((start, i + 1, cur)); cur = []; start = i + 2
seg.append((start, e, cur))
tot = {}
for a, b, c in seg:
    def n(f): return sum(1 for x in c if f(x))
    row = dict(valu=n(lambda x: x.startswith('v_')), f64=n(lambda x: re.match(r'v_\w*f64', x)), cnd=n(lambda x: x.startswith('v_cndmask')),
               mov=n(lambda x: x.startswith('v_mov') or x.startswith('v_accvgpr')), rdln=n(lambda x: x.startswith('v_readlane') or x.startswith('v_writelane') or x.startswith('v_readfirstlane')),
               ds=n(lambda x: x.startswith('ds_')), scratch=n(lambda x: x.startswith('scratch_')), salu=n(lambda x: x.startswith('s_')))
    print(f"{a:6d}-{b:6d} " + " ".join(f"{k}={v:5d}" for k, v in row.items()))
