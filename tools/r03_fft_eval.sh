#!/bin/bash
# Dev tool (GPU box): Fourier-stage evaluation of the current build: parity tests, per-class durations with the classes
# serialised on one stream, the default bench line.   usage: tools/r03_fft_eval.sh <tag> [pytest-selector]
export TMPDIR=/tmp
R=$PWD
TAG=${1:-eval}
O=$R/gpurun_out/r03_$TAG
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_trans.py -m gpu -x -q ${2:+-k "$2"} > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 -c "
import json,sys
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('value', round(d['value'],2), 'ms/step', round(d['ms_per_step'],3), [(k['name'][:24], round(k['avg_ms'],3)) for k in d.get('roofline_kernels',[])])
"
cd /tmp
ATLAS_AMD_FFT_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/serial --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/serial.log 2>&1
cd $R
python3 - << PY > $O/classes.txt 2>&1
import csv, glob, collections
dur = collections.defaultdict(list)
for f in glob.glob('$O/serial/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0][-48:]
        dur[k].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6)
tot = 0
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if 'fft' in k or 'legendre_kernel' in k:
        print('%-50s %4d %8.3f' % (k, len(v), sum(v) / len(v)))
        if 'fft' in k: tot += sum(v) / 6.0
print('fft sum per transform, classes serialised:', round(tot, 3))
PY
head -12 $O/classes.txt; tail -1 $O/classes.txt
