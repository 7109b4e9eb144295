"""instruction totals of one loop of a kernel in a hipcc -S listing (every block between the loop header label and the last
branch back to it counted once):   python tools/asm_loop.py file.s mangled-substring [header-label]
without a header label: the loop with the most instructions"""
import re, sys, collections
s = open(sys.argv[1]).read()
m = re.search(r'^(_Z\S*' + re.escape(sys.argv[2]) + r'\S*):[^\n]*\n(.*?)\n\s*s_endpgm', s, re.S | re.M)
body = m.group(2).split('\n')
pos = {re.match(r'^(\.LBB\d+_\d+):', l).group(1): i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)}
loops = []
for i, l in enumerate(body):
    mm = re.match(r'\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.match(r'\s*s_branch\s+(\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in pos and pos[mm.group(1)] < i:
        loops.append((pos[mm.group(1)], i, mm.group(1)))
if len(sys.argv) > 3:
    loops = [x for x in loops if x[2] == sys.argv[3]]
cost = {'v_cndmask': 6, 'v_fma_f64': 4.6, 'v_fmac_f64': 4.6, 'v_mul_f64': 4.6, 'v_add_f64': 4.6, 'v_max_f64': 4.7, 'v_cmp': 4.7, 'v_rcp_f64': 9,
        'v_rndne_f64': 4.6, 'v_ldexp_f64': 4.6, 'v_cvt': 4.6, 'v_permlane': 4.6, 'v_mfma_f64_4x4x4': 18, 'v_mfma_f64_16x16x4': 66}
for a, b, lab in sorted(set(loops), key=lambda x: x[0] - x[1])[:int(sys.argv[4]) if len(sys.argv) > 4 else 1]:
    c = collections.Counter()
    clk = 0.0
    for l in body[a:b + 1]:
        l = l.strip()
        if not l or l.startswith(';') or l.startswith('.'): continue
        op = l.split()[0]
        c[op] += 1
        if op.startswith('s_nop'):
            clk += int(l.split()[1]) + 1
        elif op.startswith('v_'):
            clk += next((v for k, v in cost.items() if op.startswith(k)), 2.3)
    tot = sum(c.values())
    valu = sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
    print('loop %s: %d instructions, %d VALU, %d MFMA, %d s_nop, %d global, %d ds, %d salu; modelled SIMD clocks %.0f' % (
        lab, tot, valu, sum(v for k, v in c.items() if k.startswith('v_mfma')), c['s_nop'], sum(v for k, v in c.items() if k.startswith('global')),
        sum(v for k, v in c.items() if k.startswith('ds_')), sum(v for k, v in c.items() if k.startswith('s_') and not k.startswith('s_nop')), clk))
    print('   ' + ', '.join('%d %s' % (v, k) for k, v in c.most_common(45)))
