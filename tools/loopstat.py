"""Instruction mix of the loops that contain a barrier in one kernel of a hipcc -S listing (the producer / consumer loops of k_rollout_pc):
python tools/loopstat.py <file.s> [regex of the kernel symbol line].  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DPTG_PART=1 --cuda-device-only -S ..."""
import re,collections,sys
lines=open(sys.argv[1]).read().split('\n')
pat=sys.argv[2] if len(sys.argv)>2 else r'^_ZN12_GLOBAL__N_112k_rollout_pcILi0ELb1ELi2ELb1ELb1EfLb0E.*:'
start=None
for i,l in enumerate(lines):
    if re.match(pat, l): start=i; break
end=None
for i in range(start, len(lines)):
    if lines[i].strip().startswith('s_endpgm'): end=i; break
body=lines[start:end+1]
labels={}
for i,l in enumerate(body):
    m=re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)]=i
loops=[]
for i,l in enumerate(body):
    m=re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l) or re.search(r's_branch (\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)]<i:
        loops.append((labels[m.group(1)], i))
def stat(seg):
    c=collections.Counter()
    for l in seg:
        if not l.startswith('\t'): continue
        t=l.strip()
        if t.startswith(('.',';')): continue
        op=t.split()[0]
        if op.startswith('s_nop'): c['s_nop']+=1
        elif 'saveexec' in op or (op.startswith('s_') and 'exec' in t): c['execmask']+=1
        elif op.startswith('s_cbranch') or op.startswith('s_branch'): c['branch']+=1
        elif op.startswith('s_waitcnt'): c['waitcnt']+=1
        elif op.startswith('s_barrier'): c['barrier']+=1
        elif op.startswith('s_'): c['salu']+=1
        elif op.startswith('v_'): c['valu']+=1
        elif op.startswith('ds_'): c['ds']+=1
        else: c['vmem']+=1
    return c
seen=set()
for a,b in sorted(loops, key=lambda x:(x[0],-x[1])):
    seg=body[a:b+1]
    c=stat(seg)
    if c['barrier']==0: continue
    key=(a)
    if key in seen: continue
    seen.add(key)
    print('loop lines %d-%d: total %d %s' % (a,b,sum(c.values()),dict(c)))
