"""Static look at a kernel's ISA: instruction classes per loop that contains MFMAs, and where the spills are.
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only openmpl_amd/csrc/h2_gemm.hip -o /tmp/h2_gemm.s
   python tools/isa_loops.py /tmp/h2_gemm.s <first line> <last line>      (the line range of one kernel: grep -n s_endpgm)
Round 6: the 223 SGPR spills of h2_stack_kernel<2> and the 47 scratch accesses of h2_stack2_kernel<2> that the round-5 review
listed are v_readlane / v_writelane / scratch_* at phase boundaries and in the straight-line tails; the steady-state k loops
(e.g. 96 instructions for two stages: 24 MFMA, 20 ds_read_b128, 4 LDS-DMA pieces, 30 SALU, 10 s_waitcnt) contain none."""
import re,sys
src=open(sys.argv[1]).read().split('\n')
start=int(sys.argv[2]); end=int(sys.argv[3])
lines=src[start:end]
labels={}
for i,l in enumerate(lines):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: labels[m.group(1)]=i
def cls(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('ds_read') or op.startswith('ds_load'): return 'ds_read'
    if op.startswith('ds_'): return 'ds_write'
    if 'load_lds' in op or ('lds' in op and op.startswith('buffer_load')): return 'dma'
    if op.startswith('global_load') or op.startswith('buffer_load'): return 'vload'
    if op.startswith('global_store') or op.startswith('buffer_store'): return 'vstore'
    if op.startswith('v_readlane') or op.startswith('v_writelane'): return 'spill_lane'
    if op.startswith('scratch_'): return 'scratch'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'branch'
    if op.startswith('s_'): return 'salu'
    return 'other'
loops=[]
for i,l in enumerate(lines):
    m=re.match(r'^\s+(s_cbranch\w+|s_branch)\s+(\.LBB\d+_\d+)',l)
    if m and m.group(2) in labels and labels[m.group(2)]<i:
        loops.append((labels[m.group(2)],i,m.group(2)))
for a,b,lab in loops:
    h={}
    for l in lines[a:b+1]:
        m=re.match(r'^\s+([a-z_0-9]+)',l)
        if not m: continue
        op=m.group(1)
        if op.startswith('.') : continue
        c=cls(op); h[c]=h.get(c,0)+1
    n=sum(h.values())
    if h.get('mfma',0)>=10:
        print(lab, 'lines',start+a,start+b,'insts',n, dict(sorted(h.items())))
# spills: scratch_* and v_readlane / v_writelane by innermost enclosing loop
def inner(i):
    best=None
    for a,b,_ in loops:
        if a<=i<=b and (best is None or (b-a)<(best[1]-best[0])): best=(a,b)
    return best
from collections import Counter
c=Counter()
for i,l in enumerate(lines):
    if re.match(r'^\s+(scratch_|v_readlane|v_writelane)',l):
        k=inner(i)
        c[(k, 'scratch' if 'scratch_' in l else 'lane')]+=1
print('spill instructions by innermost loop (None = outside every loop):')
for (k,kind),v in sorted(c.items(), key=lambda kv:-kv[1])[:12]:
    if k:
        n_mfma=sum(1 for l in lines[k[0]:k[1]] if 'v_mfma' in l)
        print('  lines %d-%d (%d instructions, %d MFMA): %d %s'%(start+k[0],start+k[1],k[1]-k[0],n_mfma,v,kind))
    else: print('  outside loops: %d %s'%(v,kind))
