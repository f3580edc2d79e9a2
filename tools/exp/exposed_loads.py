#!/usr/bin/env python3
"""Loads that the generated code waits for ON THE SPOT (global_load ... s_waitcnt vmcnt(0) within three instructions) in the hot kernels
of the fused engine: each is a memory round trip nothing else hides.  Works without a GPU:
    cd opticommpy_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I. --cuda-device-only -S engine_fused_f64.hip -o /tmp/f64.s
    python tools/exp/exposed_loads.py /tmp/f64.s [/tmp/f32.s]"""
import re,sys,subprocess
def demangle(n):
    return subprocess.run(['c++filt',n],capture_output=True,text=True).stdout.strip().replace('ssf::(anonymous namespace)::','').split('(ssf::')[0].replace('void ','')
for f in sys.argv[1:]:
    lines=open(f).read().split('\n')
    name=None; body=[]
    kernels={}
    for ln in lines:
        m=re.match(r'^(_ZN3ssf\S+):\s*; @',ln)
        if m: name=m.group(1); kernels[name]=[]; continue
        if name is not None:
            t=ln.strip()
            if t and not t.startswith(';') and not t.startswith('.'): kernels[name].append(t.split(';')[0].strip())
            if t.startswith('s_endpgm'): name=None
    for k,ins in kernels.items():
        d=demangle(k)
        if not re.search(r'k_col<double, 8, 3, 0, (2|4|9)>|k_col_pk<10, 0, (2|4|9)>|k_row<(double|float __vector\(2\)), 256, 2, 12>',d): continue
        exposed=[]
        for i,t in enumerate(ins):
            if t.startswith('global_load') or t.startswith('buffer_load'):
                for j in range(i+1,min(i+4,len(ins))):
                    if ins[j].startswith('s_waitcnt') and 'vmcnt(0)' in ins[j]:
                        exposed.append(i); break
                    if ins[j].startswith('global_load') or ins[j].startswith('buffer_load'): break
        print(d, 'instructions', len(ins), 'loads waited for on the spot:', len(exposed), exposed[:24])
