"""Instruction mix per basic block of one kernel in a hipcc -S listing.  usage: isa_blocks.py file.s symbol-prefix"""
import sys, collections
lines = open(sys.argv[1]).read().split('\n')
pre = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith(pre) and ':' in l.split()[0]][0]
end = [i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm')][0]
def kind(op):
    if 'mfma' in op: return 'mfma'
    if op.startswith('ds_read'): return 'dsr'
    if op.startswith('ds_write'): return 'dsw'
    if op.startswith('global_load'): return 'gld'
    if op.startswith('global_store'): return 'gst'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'bar'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'br'
    if op.startswith('s_'): return 's'
    if op.startswith('v_'): return 'v'
    return op
tot = collections.Counter(); cur = collections.Counter(); name = 'entry'; out = []
for l in lines[start + 1:end + 1]:
    s = l.strip()
    if s.startswith('.LBB') and ':' in s.split()[0]:
        out.append((name, cur)); cur = collections.Counter(); name = s; continue
    t = s.split()
    if not t or t[0].startswith(('.', ';')): continue
    cur[kind(t[0])] += 1; tot[kind(t[0])] += 1
out.append((name, cur))
print('total', dict(tot))
for n, c in out:
    if sum(c.values()) >= int(sys.argv[3]) if len(sys.argv) > 3 else 8: print(n, dict(c))
