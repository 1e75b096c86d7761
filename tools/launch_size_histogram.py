#!/usr/bin/env python3
"""Launch-size histogram of one factorisation from a rocprofv3 kernel trace: launches by workgroup count, with durations.

    python tools/launch_size_histogram.py <kernel_trace.csv>
"""
import csv,sys,collections
f=sys.argv[1]
rows=list(csv.DictReader(open(f)))
def short(n):
    n=n.replace("(anonymous namespace)::","").split("(")[0]
    return n[5:] if n.startswith("void ") else n
rows=[r for r in rows if 'rocclr' not in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find the last full factorisation: split at gaps > 3 ms? use getrf_tiled count 141 per factorisation
# factorisations are separated by the value reset (device idle for more than 2 ms, or copy kernels filtered out above)
segs=[[rows[0]]]
for prev,r in zip(rows[:-1],rows[1:]):
    if int(r['Start_Timestamp'])-int(prev['End_Timestamp'])>2_000_000:
        segs.append([])
    segs[-1].append(r)
segs=[x for x in segs if len(x)>200]
print('factorisations found',len(segs),[len(x) for x in segs])
seg=segs[-2] if len(segs)>1 else segs[-1]
print('span ms',(int(seg[-1]['End_Timestamp'])-int(seg[0]['Start_Timestamp']))/1e6)
for key in ('ssssm_tiles','ssssm_front','ssssm_dense','trsm_dense_direct','getrf_tiled','densify','sparsify'):
    b=collections.defaultdict(lambda:[0,0.0])
    for r in seg:
        n=short(r['Kernel_Name'])
        if not n.startswith(key): continue
        wg=int(r.get('Grid_Size_X') or r.get('Grid_Size'))//int(r.get('Workgroup_Size_X') or r.get('Workgroup_Size'))
        d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
        k=1
        while k<wg: k*=2
        b[k][0]+=1; b[k][1]+=d
    print(key)
    for k in sorted(b): print('   wgs<=%6d: %3d launches, total %8.1f us, avg %7.1f us'%(k,b[k][0],b[k][1],b[k][1]/b[k][0]))

# what else was on the device while each GETRF launch ran
print('getrf launches: workgroups, duration us, start offset ms, overlapping kernels (us of overlap)')
t0=int(seg[0]['Start_Timestamp'])
for r in seg:
    n=short(r['Kernel_Name'])
    if not n.startswith('getrf'): continue
    a,b=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    ov=collections.defaultdict(float)
    for q in seg:
        if q is r: continue
        c,d=int(q['Start_Timestamp']),int(q['End_Timestamp'])
        o=min(b,d)-max(a,c)
        if o>0: ov[short(q['Kernel_Name'])[:18]]+=o/1e3
    wg=int(r.get('Grid_Size_X') or r.get('Grid_Size'))//int(r.get('Workgroup_Size_X') or r.get('Workgroup_Size'))
    print('  %4d wgs %7.1f us at %6.2f ms | %s'%(wg,(b-a)/1e3,(a-t0)/1e6,', '.join('%s %.0f'%(k,v) for k,v in sorted(ov.items(),key=lambda x:-x[1]))))
