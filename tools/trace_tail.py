#!/usr/bin/env python3
"""The steady state of the streaming pipeline in a rocprofv3 kernel trace: the window from the
start of the N-th-last forward launch to the last kernel's end (the last pass of
tools/gpu_inflate_split.py --repeat 2 is 32 forward launches) - how long it is, for how much of it
any kernel / each kind of kernel is on the GPU, average durations.
Usage: python tools/trace_tail.py DIR_WITH_kernel_trace.csv [N=32] [list]"""
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
ev=[]
for r in rows:
    n=r['Kernel_Name']
    k='k1' if 'tokens' in n else 'k2' if 'resolve' in n else 'fwd' if 'forward' in n else 'merge' if 'merge' in n else 'combine' if 'combine' in n else 'fill' if 'fill' in n else n[:20]
    ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),k,r.get('Queue_Id','?')))
ev.sort()
fw=[e for e in ev if e[2]=='fwd']
n_last=int(sys.argv[2]) if len(sys.argv)>2 else 32
t_from=fw[-n_last][0]-1; t_to=max(e[1] for e in ev)
sel=[e for e in ev if e[0]>=t_from]
print('window ms',(t_to-t_from)/1e6,'kernels',len(sel))
def union(v):
    v=sorted((s,e) for s,e,_,_ in v)
    if not v: return 0
    busy=0; cs,ce=v[0]
    for s,e in v[1:]:
        if s>ce: busy+=ce-cs; cs,ce=s,e
        else: ce=max(ce,e)
    return (busy+ce-cs)/1e6
print('any kernel busy ms',round(union(sel),1))
for k in ('fwd','k1','k2','merge','combine','fill'):
    v=[e for e in sel if e[2]==k]
    if v: print(k,'n',len(v),'avg ms',round(sum(e[1]-e[0] for e in v)/len(v)/1e6,2),'sum ms',round(sum(e[1]-e[0] for e in v)/1e6,1),'union ms',round(union(v),1))
if len(sys.argv)>3:
    for s,e,k,q in sel[:120]: print(round((s-t_from)/1e6,2),round((e-t_from)/1e6,2),k,q)
