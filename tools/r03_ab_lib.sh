#!/bin/bash
# Dev tool (GPU box): A/B of a dev build of the library against the product build, alternating, Fourier stage times
# usage: tools/r03_ab_lib.sh atlas_amd/lib/dev/libatlas_amd_<variant>.so [streams]
V=$1; S=${2:-4}
for rep in 1 2 3; do for lib in "" "$V"; do
  if [ -z "$lib" ]; then unset ATLAS_AMD_LIB; tag=product; else export ATLAS_AMD_LIB=$lib; tag=variant; fi
  ATLAS_AMD_FFT_STREAMS=$S python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep $tag streams $S', 'ms/step', round(d['ms_per_step'],3), [round(k['avg_ms'],3) for k in d.get('roofline_kernels',[])])"
done; done
