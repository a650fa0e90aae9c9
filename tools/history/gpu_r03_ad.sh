#!/bin/bash
# visit AD: kernel composition of a LoLa chain after the dispatch cuts
O=gpurun_out/r03ad; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o lola -- python $GRAFT_REPO_ROOT/bench.py --workload lola --steps 6 --warmup 2 --no-unchanged-caller > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, collections, glob
f=glob.glob('gpurun_out/r03ad/prof/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
qs=collections.Counter(r['Queue_Id'] for r in rows)
q0=qs.most_common(1)[0][0]
q=[r for r in rows if r['Queue_Id']==q0]
q.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in q]
idx=[i for i,n in enumerate(names) if 'k_square_fused' in n]
starts=idx[::4]
seg=q[starts[-3]:starts[-2]]
c=collections.Counter(); d=collections.Counter()
for r in seg:
    key=(r['Kernel_Name'].split('(')[0].replace('void ','')[:50], r['Grid_Size_X'])
    c[key]+=1; d[key]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
span=(int(seg[-1]['End_Timestamp'])-int(seg[0]['Start_Timestamp']))/1e3
print("one image on one queue: %d dispatches, busy %.0f us, span %.0f us" % (len(seg), sum(d.values()), span))
for k,v in sorted(c.items(), key=lambda kv:-d[kv[0]])[:22]: print("%-50s grid %8s x%3d  total %7.1f us  avg %6.1f" % (k[0], k[1], v, d[k], d[k]/v))
PY
