#!/bin/bash
# round 3, sixth GPU job: where do the update kernels' bytes come from?  PMC passes over the stand-alone front benchmark
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for cfg in "40 1 100" "40 1 45"; do
tag=$(echo $cfg | tr ' ' '_')
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum"; do
  name=$(echo $pmc | tr ' ' '+' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $R/gpurun_out/prof_r03f_$tag/$name -o run -- $R/tools/microbench/front_gemm.bin $cfg > /dev/null 2>&1
done
python3 - $R/gpurun_out/prof_r03f_$tag <<'PY' > $R/gpurun_out/r03f_pmc_$tag.txt
import csv,glob,sys,collections,os
root=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root,'*','*counter_collection.csv'))+glob.glob(os.path.join(root,'*','*','*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].split('(')[0].replace('void ','')
        agg[n][r['Counter_Name']].append((int(r['Dispatch_Id']),float(r['Counter_Value'])))
for n in sorted(agg):
    if 'ssssm' not in n: continue
    print(n)
    for c in sorted(agg[n]):
        v=agg[n][c]
        # the timing section launches each kernel 5 x 2 times: mean over the dispatches of the last half
        vals=[x for _,x in sorted(v)]
        tail=vals[len(vals)//2:]
        print("   %-32s dispatches %3d  mean of last half %.4g" % (c,len(vals),sum(tail)/len(tail)))
PY
cat $R/gpurun_out/r03f_pmc_$tag.txt
find $R/gpurun_out/prof_r03f_$tag -name "*kernel_trace.csv" -delete
done
