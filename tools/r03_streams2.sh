#!/bin/bash
# Dev tool (GPU box): Fourier stage time vs number of class streams, three repeats each
for rep in 1 2 3; do for s in 4 5 6 8; do
  ATLAS_AMD_FFT_STREAMS=$s python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep streams $s', 'ms/step', round(d['ms_per_step'],3), [round(k['avg_ms'],3) for k in d.get('roofline_kernels',[])])"
done; done
