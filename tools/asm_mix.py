"""Instruction mix of the largest basic blocks of one kernel in a hipcc -S listing:  python tools/asm_mix.py file.s name_substring [n]"""
import collections, re, sys
src = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 2
start = [i for i, l in enumerate(src) if l.startswith('_Z') and key in l and ':' in l.split(';')[0]][0]
end = [i for i in range(start, len(src)) if src[i].strip().startswith('.Lfunc_end')][0]
blocks, cur = [], None
for i in range(start, end):
    l = src[i]
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        cur = [m.group(1), collections.Counter()]
        blocks.append(cur)
        continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    if cur:
        cur[1][t.split()[0]] += 1
for name, c in sorted(blocks, key=lambda b: -sum(b[1].values()))[:top]:
    valu = sum(n for op, n in c.items() if op.startswith('v_') and not op.startswith('v_mfma') and not op.startswith('v_exp'))
    print(name, 'instructions', sum(c.values()), '| valu', valu, '| exp', c['v_exp_f32_e32'], '| mfma',
          sum(n for op, n in c.items() if op.startswith('v_mfma')), '| lds', sum(n for op, n in c.items() if op.startswith('ds_')))
    print('   ', c.most_common(20))
