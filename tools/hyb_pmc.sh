#!/bin/bash
# Dev tool (GPU box): SQ counters of the Fourier kernels with the hybrid rows enabled (counters only).
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/hybpmc
rm -rf $O; mkdir -p $O
cd /tmp
i=0
for pass in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS" \
  ; do
  i=$((i+1))
  ATLAS_AMD_FFT_HYBRID=1 rocprofv3 --kernel-trace --pmc $pass -d $O/p$i --output-format csv -- timeout 200 python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/p$i.log 2>&1
done
cd $R
python3 - << 'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob('gpurun_out/hybpmc/p*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'fft_rows' not in k: continue
        k = k.split('(')[0][-44:]
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); disp[(k, row['Counter_Name'])].add(row['Dispatch_Id'])
for k in agg:
    if 'hyb' not in k and '5, 10' not in k and '1, 12' not in k: continue
    print(k)
    for c, v in sorted(agg[k].items()):
        print('    %-36s total %.5g   (%d dispatches)' % (c, v, len(disp[(k, c)])))
PY
