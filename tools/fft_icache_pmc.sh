#!/bin/bash
# Dev tool (GPU box) [r4]: instruction-cache counters of the Fourier kernels (counters only; rocprofv3 --pmc with --kernel-trace):
# requests / hits / misses of the 64-KB instruction cache two CUs share, and the fetches in flight, once with the product's four
# streams (row classes of different kernels side by side on a CU pair) and once with the classes serialised on one stream.
#   tools/fft_icache_pmc.sh [outdir-tag]        extra environment (e.g. ATLAS_AMD_FFT_NATIVE=1) is passed through
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-icache}
rm -rf $O; mkdir -p $O
cd /tmp
for streams in 4 1; do
  i=0
  for pass in \
    "SQ_WAVES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
    "SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
    ; do
    i=$((i+1))
    ATLAS_AMD_FFT_STREAMS=$streams timeout 300 rocprofv3 --kernel-trace --pmc $pass -d $O/s${streams}_p$i --output-format csv -- \
      python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/s${streams}_p$i.log 2>&1
  done
done
cd $R
python3 - "$O" << 'PY' > $O/summary.txt 2>&1
import csv, glob, collections, sys
O = sys.argv[1]
for streams in (4, 1):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for f in glob.glob('%s/s%d_p*/**/*counter_collection.csv' % (O, streams), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name']
            if 'fft_rows' not in k and 'legendre_kernel' not in k and 'legendre_lean' not in k: continue
            k = k.split('(')[0][-56:]
            agg[k][row['Counter_Name']] += float(row['Counter_Value']); disp[(k, row['Counter_Name'])].add(row['Dispatch_Id'])
    print('== ATLAS_AMD_FFT_STREAMS=%d: per dispatch' % streams)
    tot = collections.defaultdict(float)
    for k in sorted(agg, key=lambda k: -agg[k].get('SQ_WAVE_CYCLES', 0)):
        c = {n: v / len(disp[(k, n)]) for n, v in agg[k].items()}
        req, hit, mis = c.get('SQC_ICACHE_REQ', 0), c.get('SQC_ICACHE_HITS', 0), c.get('SQC_ICACHE_MISSES', 0)
        dup = c.get('SQC_ICACHE_MISSES_DUPLICATE', 0)
        wc = c.get('SQ_WAVE_CYCLES', 0)
        print('%-58s waves %8d  icache req %.3g hits %.3g (%.1f %%) misses %.3g dup %.3g | ifetch %.3g  level/fetch %.1f | wave cycles %.3g '
              'wait-inst %.1f %% wait-any %.1f %% active %.1f %%' % (
                  k, c.get('SQ_WAVES', 0), req, hit, 100 * hit / req if req else 0, mis, dup, c.get('SQ_IFETCH', 0),
                  c.get('SQ_IFETCH_LEVEL', 0) / c['SQ_IFETCH'] if c.get('SQ_IFETCH') else 0, wc,
                  100 * c.get('SQ_WAIT_INST_ANY', 0) / wc if wc else 0, 100 * c.get('SQ_WAIT_ANY', 0) / wc if wc else 0,
                  100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc if wc else 0))
        if 'fft' in k:
            for n in ('SQC_ICACHE_REQ', 'SQC_ICACHE_HITS', 'SQC_ICACHE_MISSES', 'SQC_ICACHE_MISSES_DUPLICATE'): tot[n] += c.get(n, 0)
    if tot['SQC_ICACHE_REQ']:
        print('   Fourier kernels together: hits %.2f %% of %.4g requests, misses %.4g (+ %.4g duplicates)' % (
            100 * tot['SQC_ICACHE_HITS'] / tot['SQC_ICACHE_REQ'], tot['SQC_ICACHE_REQ'], tot['SQC_ICACHE_MISSES'], tot['SQC_ICACHE_MISSES_DUPLICATE']))
PY
cat $O/summary.txt
