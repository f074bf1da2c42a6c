#!/usr/bin/env python3
"""Outline of a kernel's ISA: labels, branches, barriers, scratch traffic and MFMA counts in program order.
   hipcc ... -S --cuda-device-only x.hip -o x.s ; python tools/asm_outline.py x.s [name-substring]"""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
want = sys.argv[2] if len(sys.argv) > 2 else ''
cur, ev, mf = None, [], 0
def flush():
    global ev, mf
    if cur is None: return
    if mf: ev.append(f'mfma x{mf}')
    out = []
    for e in ev:
        if out and out[-1][0] == e: out[-1][1] += 1
        else: out.append([e, 1])
    print(cur)
    print('  ' + ' '.join(f'{e}*{n}' if n > 1 else e for e, n in out))
for ln in txt:
    t = ln.strip()
    m = re.match(r'^(_Z\w+):', ln)
    if m:
        flush(); cur, ev, mf = (m.group(1) if want in m.group(1) else None), [], 0
        continue
    if cur is None: continue
    if t.startswith('.Lfunc_end'):
        flush(); cur = None; continue
    if t.startswith('v_mfma'): mf += 1; continue
    key = None
    if t.startswith('s_barrier'): key = 'BARRIER'
    elif t.startswith('scratch_load'): key = 'sL'
    elif t.startswith('scratch_store'): key = 'sS'
    elif t.startswith('s_cbranch') or t.startswith('s_branch'): key = t.split()[0][2:] + '->' + t.split()[-1]
    elif re.match(r'^\.LBB\d+_\d+:', t): key = t.split(':')[0]
    elif t.startswith('global_load_lds'): key = 'dma'
    elif t.startswith('s_waitcnt vmcnt(0)'): key = 'VM0'
    if key:
        if mf: ev.append(f'mfma x{mf}'); mf = 0
        ev.append(key)
flush()
