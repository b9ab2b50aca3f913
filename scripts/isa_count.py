import re,collections,sys
pat=sys.argv[1]
lines=open('/tmp/tc.s').read().splitlines()
start=None
for i,l in enumerate(lines):
    if l.startswith(pat) and ':' in l:
        start=i; break
ops=collections.Counter()
for l in lines[start+1:]:
    t=l.strip()
    if t.startswith('.Lfunc_end'): break
    if not t or t.startswith(('.',';')) or t.split()[0].endswith(':'): continue
    ops[t.split()[0]]+=1
cls=collections.Counter()
for op,c in ops.items():
    k='valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_','buffer_','scratch_','flat_')) else 'other'
    cls[k]+=c
print(sum(ops.values()), dict(cls)); print(ops.most_common(36))
