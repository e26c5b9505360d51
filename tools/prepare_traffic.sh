#!/bin/bash
# Dev tool (GPU box) [r5]: HBM traffic per launch of the vor/div preparation kernel in its two forms (ATLAS_AMD_PREPARE=rows|stream),
# TL1279 -> O1280, nscalar 137 + nvordiv 137 (tools/bench_configs.py --only C4vd); counters only, one pass per counter
# (FETCH_SIZE doubled on gfx950 as in tools/prof_round_summary.py).   tools/prepare_traffic.sh [outdir-tag]
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-prepare_traffic}
rm -rf $O; mkdir -p $O
cd /tmp
for form in rows stream; do
  for c in FETCH_SIZE WRITE_SIZE; do
    ATLAS_AMD_PREPARE=$form timeout 150 rocprofv3 --kernel-trace --pmc $c -d $O/${form}_$c --output-format csv -- \
        python $R/tools/bench_configs.py --only C4vd > $O/${form}_$c.log 2>&1
  done
done
cd $R
python3 - "$O" << 'PY'
import csv, glob, collections, sys
O = sys.argv[1]
ALG = 10.81e9   # inputs 3 x 1.80 GB + merged spectra 5.41 GB
for form in ('rows', 'stream'):
    c = {}
    for name in ('FETCH_SIZE', 'WRITE_SIZE'):
        tot, ids = 0.0, set()
        for f in glob.glob('%s/%s_%s/**/*counter_collection.csv' % (O, form, name), recursive=True):
            for row in csv.DictReader(open(f)):
                if 'spectra_prepare' in row['Kernel_Name'] and row['Counter_Name'] == name:
                    tot += float(row['Counter_Value']); ids.add(row['Dispatch_Id'])
        c[name] = tot / max(len(ids), 1)
    rd, wr = 2 * c['FETCH_SIZE'] * 1e3, c['WRITE_SIZE'] * 1e3
    print('%-7s read %.3f GB  written %.3f GB  = %.3f GB per launch = %.2f x the algorithmic 10.81 GB' % (form, rd / 1e9, wr / 1e9, (rd + wr) / 1e9, (rd + wr) / ALG))
PY
