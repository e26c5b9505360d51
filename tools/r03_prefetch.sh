#!/bin/bash
# Dev tool (GPU box): Fourier stage time by L2-prefetch setting (ATLAS_AMD_FFT_PREFETCH="distance,requests per line"), alternating
# usage: tools/r03_prefetch.sh "0 8,2 16,2 8,4" [repeats]
SET=${1:-"0 8,2"}; REP=${2:-2}
for rep in $(seq $REP); do for s in $SET; do
  ATLAS_AMD_FFT_PREFETCH=$s python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep prefetch $s', 'ms/step', round(d['ms_per_step'],3), [round(k['avg_ms'],3) for k in d.get('roofline_kernels',[])])"
done; done
