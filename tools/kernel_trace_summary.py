import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
d=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    k='k1' if 'tokens' in n else 'k2' if 'resolve' in n else 'fwd' if 'forward' in n else n[:30]
    d[k].append((int(r['Start_Timestamp']),int(r['End_Timestamp'])))
t0=min(s for v in d.values() for s,e in v); t1=max(e for v in d.values() for s,e in v)
print('span ms',(t1-t0)/1e6)
for k,v in sorted(d.items()):
    dur=[(e-s)/1e6 for s,e in v]
    # union busy time
    v=sorted(v); busy=0; cs,ce=v[0]
    for s,e in v[1:]:
        if s>ce: busy+=ce-cs; cs,ce=s,e
        else: ce=max(ce,e)
    busy+=ce-cs
    print(k,'n',len(v),'avg ms',round(sum(dur)/len(dur),2),'max',round(max(dur),2),'union busy ms',round(busy/1e6,1))
