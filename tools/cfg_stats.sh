#!/bin/bash
# Dev tool (GPU box) [r4]: per-kernel durations of one configuration of tools/bench_configs.py (rocprofv3 kernel trace):
#   tools/cfg_stats.sh C5n [outdir-tag]       extra environment is passed through (e.g. ATLAS_AMD_FFT_STREAMS=1: classes serialised)
export TMPDIR=/tmp
CFG=${1:-C5}
R=$PWD
O=$R/gpurun_out/${2:-cfgstats}_$CFG
rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace -d $O/stats --output-format csv -- python $R/tools/bench_configs.py --only $CFG --out $O/line.jsonl > $O/stats.log 2>&1
cd $R
python3 - "$O" << 'PY'
import csv, glob, collections, sys
O = sys.argv[1]
dur = collections.defaultdict(list)
for f in glob.glob(O + '/stats/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row['Kernel_Name'].split('(')[0][-64:]].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6)
n = max(len(v) for k, v in dur.items() if 'legendre_kernel' in k or 'legendre_lean' in k)
print('kernel, launches, avg ms, ms per transform (%d transforms traced)' % n)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if 'fft' in k or 'legendre' in k:
        print('%-66s %4d %8.3f %8.3f' % (k, len(v), sum(v) / len(v), sum(v) / n))
PY
