#!/bin/bash
# Dev tool (GPU box), round 3 first call: (1) fp64 VALU / LDS-stage probe, (2) baseline bench, (3) per-class Fourier kernel
# durations with the classes serialised on one stream, (4) SQ wait/issue counters of the Fourier classes (counters only).
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03c1
rm -rf $O; mkdir -p $O
timeout 120 tools/probe/probe_fft_valu_gfx950 > $O/probe_fft_valu.txt 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cd /tmp
ATLAS_AMD_FFT_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/serial --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/serial.log 2>&1
i=0
for pass in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES" \
  "SQ_WAVES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" \
  ; do
  i=$((i+1))
  ATLAS_AMD_FFT_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc $pass -d $O/pmc$i --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc$i.log 2>&1
done
cd $R
python3 - << 'PY' > gpurun_out/r03c1/summary.txt 2>&1
import csv, glob, collections
O = 'gpurun_out/r03c1'
dur = collections.defaultdict(list)
for f in glob.glob(O + '/serial/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0][-48:]
        dur[k].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6)
print('== serialised classes: kernel, launches, avg ms, total per transform')
tot = 0
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if 'fft' in k or 'legendre_kernel' in k:
        n = len(v); avg = sum(v) / n
        print('%-50s %4d %8.3f' % (k, n, avg))
        if 'fft' in k: tot += sum(v) / 6.0
print('fft sum per transform (6 transforms traced):', tot)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob(O + '/pmc*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'fft_rows' not in k: continue
        k = k.split('(')[0][-48:]
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); disp[(k, row['Counter_Name'])].add(row['Dispatch_Id'])
for k in agg:
    print(k)
    for c, v in sorted(agg[k].items()):
        print('    %-28s %.5g' % (c, v / len(disp[(k, c)])))
PY
cat $O/probe_fft_valu.txt
tail -1 $O/bench.json | cut -c1-600
head -30 $O/summary.txt
