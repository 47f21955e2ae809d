"""compact instruction-class view of the MFMA-containing basic blocks of one kernel: python scratch/isa_view.py file.s kernel-substring"""
import sys
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split('\n')
on, out = False, []
for l in lines:
    t = l.strip()
    if not on:
        if t.startswith("_ZN") and key in t and ":" in t:
            on = True
        continue
    if t.startswith('s_endpgm'):
        break
    if not t or t.startswith(';') or (t.startswith('.') and not t.startswith('.LBB')):
        continue
    if t.startswith('.LBB'):
        out.append('\n' + t.split(':')[0] + ' ')
        continue
    op = t.split()[0]
    c = ('M' if op.startswith('v_mfma') else 'L' if op.startswith('ds_read') else 'W' if op.startswith('ds_write') else
         'G' if op.startswith(('global_load', 'scratch_load')) else 'S' if op.startswith(('global_store', 'scratch_store')) else
         'w' if op == 's_waitcnt' else 'B' if op == 's_barrier' else 'j' if op.startswith(('s_cbranch', 's_branch')) else
         'n' if op == 's_nop' else 'v' if op.startswith('v_') else 's' if op.startswith('s_') else '?')
    out.append(c)
for blk in ''.join(out).split('\n'):
    if 'M' in blk:
        print(blk[:2500]); print()
