#!/bin/bash
# Dev tool (GPU box): which row-length classes should take the 256-register row_ct3 form (class streams: 4)
for fm in "" "0" "5120,4608" "5120,4608,6144" "5120,4608,4096" "5120,4608,3840"; do
  for s in 4 6; do
  if [ -z "$fm" ]; then unset ATLAS_AMD_FFT_FAST_M; else export ATLAS_AMD_FFT_FAST_M=$fm; fi
  ATLAS_AMD_FFT_STREAMS=$s python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('FAST_M=[$fm] streams $s', 'ms/step', round(d['ms_per_step'],3), [round(k['avg_ms'],3) for k in d.get('roofline_kernels',[])])"
done; done
