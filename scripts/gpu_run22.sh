#!/bin/bash
# dev: SQ counters of the Open3D filter kernels
set -x
export TMPDIR=/tmp
O=gpurun_out/r03v
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $O/pmc -o q --output-format csv -- python3 bench.py --precision plan --legs none --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline > $O/pmc.log 2>&1
python - <<'P'
import csv, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter(); seen=set()
for r in csv.DictReader(open('gpurun_out/r03v/pmc/q_counter_collection.csv')):
    k=r['Kernel_Name']
    if not any(t in k for t in ('ror_count','sor_knn')): continue
    kk=k[:40]
    agg[kk][r['Counter_Name']]+=float(r['Counter_Value'])
    if (kk,r['Dispatch_Id']) not in seen: seen.add((kk,r['Dispatch_Id'])); n[kk]+=1; agg[kk]['ns']+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for k,v in agg.items():
    print(k, 'launches', n[k], {a: round(b/n[k],1) for a,b in v.items()})
P
rm -f $O/pmc/q_counter_collection.csv $O/pmc/q_kernel_trace.csv
