"""Basic blocks of one kernel of a hipcc -S listing IN ORDER, with their instruction mix (loop bodies are the runs of blocks up to a
backward branch):  python tools/asm_blocks.py file.s name_substring"""
import collections, re, sys
src = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(src) if l.startswith('_Z') and key in l and ':' in l.split(';')[0]][0]
end = [i for i in range(start, len(src)) if src[i].strip().startswith('.Lfunc_end')][0]
cur = ['entry', collections.Counter(), []]
blocks = [cur]
for i in range(start + 1, end):
    l = src[i]
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        cur = [m.group(1), collections.Counter(), []]
        blocks.append(cur)
        continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    op = t.split()[0]
    cur[1][op] += 1
    if op.startswith('s_cbranch') or op == 's_branch':
        cur[2].append(t.split()[-1])
SLOW = ('v_cvt_pk', 'v_dot2', 'v_cndmask', 'v_lshl', 'v_lshr', 'v_perm', 'v_bfe', 'v_bfi')
for name, c, br in blocks:
    n = sum(c.values())
    if n == 0:
        continue
    valu = sum(k for op, k in c.items() if op.startswith('v_') and not op.startswith('v_mfma'))
    slow = sum(k for op, k in c.items() if op.startswith(SLOW) or op.endswith('_dpp'))
    pk = sum(k for op, k in c.items() if op.startswith('v_pk_'))
    exp = sum(k for op, k in c.items() if op.startswith(('v_exp', 'v_rcp', 'v_log', 'v_rsq', 'v_sqrt')))
    mfma = sum(k for op, k in c.items() if op.startswith('v_mfma'))
    lds = sum(k for op, k in c.items() if op.startswith('ds_'))
    vmem = sum(k for op, k in c.items() if op.startswith(('global_', 'buffer_', 'flat_')))
    salu = sum(k for op, k in c.items() if op.startswith('s_'))
    print(f"{name:12s} n {n:4d} | valu {valu:4d} (slow {slow:3d} pk {pk:3d} trans {exp:3d}) mfma {mfma:3d} lds {lds:3d} vmem {vmem:3d} salu {salu:3d} -> {' '.join(br)}")
