#!/bin/bash
# Dev tool (GPU box): PMC passes over the bench (counters only, no trace domains besides kernel-trace).
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/pmc
cd /tmp
rocprofv3 -L > $R/gpurun_out/pmc/counters_list.txt 2>&1
run() { # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmc/$name --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc/$name.log 2>&1
}
run p1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run p2 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD
run p3 FETCH_SIZE
run p4 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run p5 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
cd $R
python3 - << 'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc/p*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name'][:50]
            agg[k][row['Counter_Name']] += float(row['Counter_Value'])
        seen = collections.Counter()
        for row in csv.DictReader(open(f)):
            seen[(row['Kernel_Name'][:50], row['Dispatch_Id'])] += 1
        disp = collections.Counter(k for (k, d2) in seen)
        print('==', f)
        for k in agg:
            print(' ', k, 'dispatches', disp[k])
            for c, v in agg[k].items():
                print('      %-34s total %.4g   per-dispatch %.4g' % (c, v, v / max(disp[k], 1)))
PY
