"""instruction mix per basic block of one kernel in a hipcc -S listing:  python tools/asm_mix.py file.s mangled-substring [min-instr]"""
import re, sys, collections
s = open(sys.argv[1]).read()
key = sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 200
m = re.search(r'^(_Z\S*' + re.escape(key) + r'\S*):[^\n]*\n(.*?)\n\s*s_endpgm', s, re.S | re.M)
body = m.group(2).split('\n')
labels = [(-1, 'entry')] + [(i, l) for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)]
def stats(lines):
    c = collections.Counter()
    for l in lines:
        l = l.strip()
        if not l or l.startswith(';') or l.startswith('.'): continue
        op = l.split()[0]
        if op.startswith('v_mfma'): c['mfma'] += 1
        elif op.startswith('ds_write') or op.startswith('ds_store'): c['ds_write'] += 1
        elif op.startswith('ds_read') or op.startswith('ds_load'): c['ds_read'] += 1
        elif op.startswith('ds_'): c['ds_other'] += 1
        elif op.startswith('s_waitcnt'): c['waitcnt'] += 1
        elif op.startswith('s_nop'): c['nop'] += 1; c['nopcycles'] += int(l.split()[1]) + 1
        elif op.startswith('v_accvgpr'): c['accvgpr'] += 1
        elif op.startswith('scratch_'): c['scratch'] += 1
        elif op.startswith('global_'): c['global'] += 1
        elif op.startswith('v_'): c['valu'] += 1; c['v:' + op] += 1
        elif op.startswith('s_'): c['salu'] += 1
        else: c['other'] += 1
    return c
for k in range(len(labels)):
    a = labels[k][0] + 1; b = labels[k + 1][0] if k + 1 < len(labels) else len(body)
    c = stats(body[a:b])
    tot = sum(v for k_, v in c.items() if not k_.startswith('v:') and k_ != 'nopcycles')
    if tot > minn:
        print(labels[k][1], 'total', tot, {k_: v for k_, v in c.items() if not k_.startswith('v:')})
        print('   ', sorted([(v, k_[2:]) for k_, v in c.items() if k_.startswith('v:')], reverse=True)[:12])
