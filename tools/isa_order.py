"""One-line picture of a kernel loop's instruction ORDER (M = MFMA, r / w = LDS read / write, T = transcendental, . = other VALU, B = barrier, |…| = s_waitcnt):
    python tools/isa_order.py <assembly file> <mangled kernel name prefix> <MFMAs in the loop>
e.g. the split LSTM kernel's timestep loop (108 MFMAs): are a k block's operand reads issued ahead of the previous block's MFMAs?"""
import re, sys, textwrap
lines = open(sys.argv[1]).read().split('\n')
name, want = sys.argv[2], int(sys.argv[3])
start = next(i for i, l in enumerate(lines) if l.startswith(name) and l.split(';')[0].rstrip().endswith(':'))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
body = lines[start:end]
labels = {l.split(':')[0]: i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)}
best = None
for i, l in enumerate(body):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        seg = body[labels[m.group(1)]:i + 1]
        if sum('v_mfma' in x for x in seg) == want and (best is None or (i - labels[m.group(1)]) < (best[1] - best[0])):
            best = (labels[m.group(1)], i)
out = []
for x in body[best[0]:best[1] + 1]:
    t = x.strip().split()
    if not t or t[0].startswith(';') or t[0].startswith('.'):
        continue
    op = t[0]
    if op.startswith('v_mfma'): out.append('M')
    elif op.startswith(('ds_read', 'ds_load')): out.append('r')
    elif op.startswith(('ds_write', 'ds_store')): out.append('w')
    elif op.startswith('s_waitcnt'): out.append('|' + x.strip().split(None, 1)[1].split(';')[0].strip().replace(' ', '') + '|')
    elif op.startswith(('v_exp', 'v_rcp')): out.append('T')
    elif op.startswith('s_barrier'): out.append('B')
    elif op.startswith('v_'): out.append('.')
print("\n".join(textwrap.wrap(''.join(out), 160)))
