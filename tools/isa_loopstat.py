"""Main-loop instruction statistics of one kernel from the compiler's assembly: registers, instructions and MFMAs per loop
iteration (what decides whether the staging work hides behind the matrix pipe).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ifewshot_detection_amd/csrc -S --cuda-device-only -o conv.s fewshot_detection_amd/csrc/conv.hip
    python tools/isa_loopstat.py conv.s conv_gemm_split8_kernelILi256ELi128E [dump]
"""
import re,sys
from collections import Counter
src, pat = sys.argv[1], sys.argv[2]
s=open(src).read().split('\n')
start=None
for i,l in enumerate(s):
    if re.match(r'^_Z.*'+pat+r'.*:', l): start=i; break
end=None
for i in range(start,len(s)):
    if s[i].strip().startswith('s_endpgm'): end=i; break
for i in range(end,len(s)):
    if '.end_amdhsa_kernel' in s[i]: kend=i; break
meta=[l.strip() for l in s[end:kend] if any(k in l for k in ('next_free_vgpr','accum_offset','group_segment_fixed_size','private_segment_fixed_size','next_free_sgpr'))]
print(meta)
body=s[start:end+1]
labels={}
for i,l in enumerate(body):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: labels[m.group(1)]=i
def ins_of(lines):
    return [x.strip().split()[0] for x in lines if x.strip() and not x.strip().startswith(('.',';','//')) and not x.strip().split()[0].endswith(':')]
print('total instrs', len(ins_of(body)))
for i,l in enumerate(body):
    m=re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)',l) or re.search(r's_branch\s+(\.LBB\d+_\d+)',l)
    if m and m.group(1) in labels and labels[m.group(1)]<i:
        lo,hi=labels[m.group(1)],i
        ins=ins_of(body[lo:hi])
        c=Counter(ins)
        nm=sum(v for k,v in c.items() if 'mfma' in k)
        if nm==0: continue
        print('loop',m.group(1),lo,hi,'n_instr',len(ins),'mfma',nm, 'per mfma %.2f'%(len(ins)/nm))
        print(sorted(c.items(), key=lambda kv:-kv[1])[:40])
        if len(sys.argv)>3:
            print('\n'.join(body[lo:hi]))
