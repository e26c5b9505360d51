#!/bin/bash
# Dev tool (GPU box): Legendre kernel variants, alternating
python -m pytest tests/test_gpu_trans.py -m gpu -x -q -k "variants" 2>&1 | grep -E "passed|failed" | tail -2
for rep in 1 2 3; do for k in lean lean2; do
  ATLAS_AMD_LEG_KERNEL=$k python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep $k', 'ms/step', round(d['ms_per_step'],3), [round(x['avg_ms'],3) for x in d.get('roofline_kernels',[])])"
done; done
