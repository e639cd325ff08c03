"""Static instruction counts per basic block of a kernel in a hipcc -S listing (VALU / DPP / transcendental / SALU / LDS / VMEM / s_nop): what EXPERIMENTS.md (72)
quotes.  usage: hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only gsr_blend_sp.hip -o sp.s; python tools/isa_regions.py sp.s <first line> <last line>"""
import re,sys
def load(path,a,b):
    return open(path).read().split('\n')[a:b]
def count(L):
    c={'valu':0,'dpp':0,'trans':0,'salu':0,'lds':0,'vmem':0,'nop':0}
    for t in L:
        t=t.strip().split(';')[0].strip()
        if not t or t.endswith(':') or t.startswith('.'): continue
        m=t.split()[0]
        if m.startswith('v_'):
            c['valu']+=1
            if 'row_' in t or 'quad_perm' in t: c['dpp']+=1
            if re.match(r'v_(exp|rcp|rsq|sqrt|log)',m): c['trans']+=1
        elif m=='s_nop': c['nop']+=1
        elif m.startswith('s_'): c['salu']+=1
        elif m.startswith('ds_'): c['lds']+=1
        elif m.startswith(('global_','buffer_','scratch_','flat_')): c['vmem']+=1
    return c
def blocks(L):
    # split into basic-block labelled regions with loop depth comments
    out=[];cur=('entry',[]) 
    for l in L:
        if l.startswith('.LBB'):
            out.append(cur);cur=(l.strip(),[])
        else: cur[1].append(l)
    out.append(cur);return out
if __name__=="__main__":
    path,a,b=sys.argv[1],int(sys.argv[2]),int(sys.argv[3])
    L=load(path,a,b)
    print('total',count(L))
    for name,body in blocks(L):
        c=count(body)
        if c['valu']+c['lds']+c['vmem']>=8: print(name[:70].ljust(72),c)
