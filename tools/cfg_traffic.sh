#!/bin/bash
# Dev tool (GPU box) [r4]: HBM traffic per launch (FETCH_SIZE doubled on gfx950 + WRITE_SIZE, as tools/prof_round.sh) and vector-ALU
# instruction counts of the kernels of one tools/bench_configs.py configuration (counters only):  tools/cfg_traffic.sh C5 [outdir-tag]
export TMPDIR=/tmp
CFG=${1:-C5}
R=$PWD
O=$R/gpurun_out/${2:-cfgtraffic}_$CFG
rm -rf $O; mkdir -p $O
cd /tmp
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"; do
  set -- $pass; name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/$name --output-format csv -- python $R/tools/bench_configs.py --only $CFG > $O/$name.log 2>&1
done
cd $R
python3 - "$O" << 'PY'
import csv, glob, collections, sys
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob(O + '/*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'fft_rows' not in k and 'legendre_kernel' not in k and 'legendre_lean' not in k: continue
        k = k.split('(')[0][-64:]
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); disp[(k, row['Counter_Name'])].add(row['Dispatch_Id'])
tot_f = tot_w = 0.0
for k in sorted(agg, key=lambda k: -(agg[k].get('FETCH_SIZE', 0) + agg[k].get('WRITE_SIZE', 0))):
    c = {n: v / len(disp[(k, n)]) for n, v in agg[k].items()}
    fetch, write = 2 * c.get("FETCH_SIZE", 0) * 1e3, c.get("WRITE_SIZE", 0) * 1e3   # units of 1 KB; FETCH_SIZE doubled on gfx950 (tools/prof_round_summary.py)
    print('%-66s read %8.3f GB  written %8.3f GB  | VALU %.4g  LDS %.4g  SALU %.4g wave-instructions' % (
        k, fetch / 1e9, write / 1e9, c.get('SQ_INSTS_VALU', 0), c.get('SQ_INSTS_LDS', 0), c.get('SQ_INSTS_SALU', 0)))
    if 'fft' in k: tot_f += fetch; tot_w += write
print('Fourier kernels together: read %.3f GB + written %.3f GB = %.3f GB per transform' % (tot_f / 1e9, tot_w / 1e9, (tot_f + tot_w) / 1e9))
PY
