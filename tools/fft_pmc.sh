#!/bin/bash
# Dev tool (GPU box): PMC passes (SQ/SPI only: a TA_* pass hung rocprofv3 on this pool) restricted to one FFT size class (ATLAS_AMD_FFT_ONLY_M), counters only.
export TMPDIR=/tmp
M=${1:-5120}
R=$PWD
O=$R/gpurun_out/fftpmc
rm -rf $O; mkdir -p $O
cd /tmp
i=0
for pass in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
  ; do
  i=$((i+1))
  ATLAS_AMD_FFT_ONLY_M=$M rocprofv3 --kernel-trace --pmc $pass -d $O/p$i --output-format csv -- timeout 100 python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/p$i.log 2>&1
done
cd $R
python3 - << 'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob('gpurun_out/fftpmc/p*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'fft_rows' not in k and 'legendre' not in k: continue
        k = k.split('(')[0][-40:]
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); disp[(k, row['Counter_Name'])].add(row['Dispatch_Id'])
for k in agg:
    print(k)
    for c, v in sorted(agg[k].items()):
        print('    %-36s %.5g' % (c, v / len(disp[(k, c)])))
PY
