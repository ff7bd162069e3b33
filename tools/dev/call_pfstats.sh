#!/bin/bash
R=/root/repo; O=gpurun_out/r02/pfstats
mkdir -p $R/$O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O" -o s -- python "$R/tools/bench_prefill.py" 7b 128 prefill-only $PF_TUN > /dev/null 2> "$R/$O/err.txt"
cd $R
python - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/r02/pfstats/s_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
g=[r for r in rows if 'gemm_prefill' in r['Kernel_Name']][-128:]
kinds=['qkv','wo','w13','w2']; d=collections.defaultdict(list)
for i,r in enumerate(g): d[kinds[i%4]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
mb={'qkv':100.66,'wo':33.55,'w13':180.36,'w2':90.18}
for k in kinds: print(k, 'mean %.1f us'%(sum(d[k])/len(d[k])), 'min %.1f'%min(d[k]), ' %.2f TB/s'%(mb[k]/(sum(d[k])/len(d[k]))))
last=rows[-360:]
t=collections.defaultdict(float)
for r in last: t[r['Kernel_Name'].split('(')[0][-50:]]+= (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
for k,v in sorted(t.items(), key=lambda kv:-kv[1]): print('%-52s %8.1f us'%(k,v))
print('span', (int(last[-1]['End_Timestamp'])-int(last[0]['Start_Timestamp']))/1e3)
PY
