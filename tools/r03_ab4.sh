#!/bin/bash
# Dev tool (GPU box): "<lib tag or product>:<ENV=..,ENV=..>" cases, alternating: tools/r03_ab4.sh "product: late:ATLAS_AMD_FFT_PREFETCH=3" [repeats]
REP=${2:-2}
for rep in $(seq $REP); do for c in $1; do
  v=${c%%:*}; e=${c#*:}; e=${e//,/ }
  if [ $v = product ]; then unset ATLAS_AMD_LIB; else export ATLAS_AMD_LIB=$PWD/atlas_amd/lib/dev/libatlas_amd_$v.so; fi
  env $e python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep $c', 'ms/step', round(d['ms_per_step'],3), [round(k['avg_ms'],3) for k in d.get('roofline_kernels',[])])"
done; done
