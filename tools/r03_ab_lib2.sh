#!/bin/bash
# Dev tool (GPU box): like r03_ab_lib.sh with an environment for both sides: tools/r03_ab_lib2.sh <variant.so> "<ENV=..>" [repeats]
V=$1; E=$2; REP=${3:-3}
for rep in $(seq $REP); do for lib in "" "$V"; do
  if [ -z "$lib" ]; then unset ATLAS_AMD_LIB; tag=product; else export ATLAS_AMD_LIB=$lib; tag=variant; fi
  env $E python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep $tag $E', 'ms/step', round(d['ms_per_step'],3), [round(k['avg_ms'],3) for k in d.get('roofline_kernels',[])])"
done; done
