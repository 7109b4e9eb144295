"""print the instruction-class sequence of one basic block: python tools/asm_seq.py file.s mangled-substring label
M mfma, v valu, a accvgpr move, w/r ds write/read, | s_waitcnt, digits s_nop cycles, S scratch, G global, s salu"""
import re, sys
s = open(sys.argv[1]).read()
m = re.search(r'^(_Z\S*' + re.escape(sys.argv[2]) + r'\S*):[^\n]*\n(.*?)\n\s*s_endpgm', s, re.S | re.M)
body = m.group(2).split('\n')
a = [i for i, l in enumerate(body) if l.startswith(sys.argv[3] + ':')][0]
seq = ''
for l in body[a + 1:]:
    l = l.strip()
    if re.match(r'^\.LBB', l): break
    if not l or l.startswith(';') or l.startswith('.'): continue
    op = l.split()[0]
    if op.startswith('v_mfma'): seq += 'M'
    elif op.startswith('ds_write'): seq += 'w'
    elif op.startswith('ds_read'): seq += 'r'
    elif op.startswith('s_waitcnt'): seq += '|'
    elif op.startswith('s_nop'): seq += str(min(9, int(l.split()[1]) + 1))
    elif op.startswith('v_accvgpr'): seq += 'a'
    elif op.startswith('scratch'): seq += 'S'
    elif op.startswith('global'): seq += 'G'
    elif op.startswith('v_'): seq += 'v'
    elif op.startswith('s_'): seq += 's'
    else: seq += '?'
for i in range(0, len(seq), 150): print(seq[i:i + 150])
