#!/bin/bash
# Dev tool (GPU box): Fourier stage time vs number of class streams
for s in 1 2 3 4 6; do
  ATLAS_AMD_FFT_STREAMS=$s python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('streams $s', 'ms/step', round(d['ms_per_step'],3), [round(k['avg_ms'],3) for k in d.get('roofline_kernels',[])])"
done
