"""usage: tools/isa_trace.py file.s kernel_substr [first_instr]  -- waits, memory ops and branches of a kernel with source lines"""
import sys
lines = open(sys.argv[1]).read().split('\n')
sub = sys.argv[2]
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
start = None
for i, l in enumerate(lines):
    if l.startswith('_Z') and sub in l and l.rstrip().endswith(':') or (l.startswith('_Z') and sub in l and ': ' in l):
        start = i
        break
cur = 0
n = 0
for l in lines[start:]:
    t = l.strip()
    if t.startswith('.loc'):
        cur = int(t.split()[2])
        continue
    if t.startswith('s_endpgm'):
        break
    if not t or t[0] == ';':
        continue
    if t.startswith('.') and not t.endswith(':'):
        continue
    n += 1
    if n < first:
        continue
    if t.startswith(('s_waitcnt', 'global_', 's_load', 's_barrier', 'buffer_', 'scratch_', 's_cbranch', 's_branch', 's_sleep')) or (
            t.endswith(':') and t.startswith('.LBB')):
        print(n, cur, t.split(';')[0])
