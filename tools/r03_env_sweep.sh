#!/bin/bash
# Dev tool (GPU box): stage times of bench.py under several environments, alternating
# usage: tools/r03_env_sweep.sh "A=1;ATLAS_AMD_FFT_ROW_AFFINITY=1;ATLAS_AMD_FFT_ROW_AFFINITY=1 ATLAS_AMD_FFT_PREFETCH=8,1" [repeats]
IFS=';' read -ra SETS <<< "$1"; REP=${2:-2}
for rep in $(seq $REP); do for s in "${SETS[@]}"; do
  env $s python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep [$s]', 'ms/step', round(d['ms_per_step'],3), [round(k['avg_ms'],3) for k in d.get('roofline_kernels',[])])"
done; done
