#!/bin/bash
# Dev tool (GPU box): per-class Fourier kernel durations, classes serialised on one stream, for the current environment
# usage: tools/fft_classes.sh <tag>
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/fft_classes_$1; rm -rf $O; mkdir -p $O
cd /tmp
ATLAS_AMD_FFT_STREAMS=1 timeout 300 rocprofv3 --kernel-trace -d $O/serial --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/serial.log 2>&1
cd $R
python3 - << PY
import csv, glob, collections, re
dur = collections.defaultdict(list)
for f in glob.glob('$O/serial/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'fft_rows' in k:
            m = re.search(r'CtShape<(\d+), (\d+)>', k)
            key = ('M=%d' % (int(m.group(1)) << int(m.group(2)))) + (' dct' if 'dct' in k else '') if m else 'generic'
            dur[key].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6)
tot = sum(sum(v) for v in dur.values()) / 6.0
print('$1', 'sum %.3f ms;' % tot, '  '.join('%s %.3f' % (k, sum(v) / 6.0) for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:12]))
PY
