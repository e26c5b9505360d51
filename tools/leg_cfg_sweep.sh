#!/bin/bash
# Dev tool (GPU box): Legendre tiling sweep, ATLAS_AMD_LEG_CFG="rtw,nrg"
for c in ${@:-"3,2" "3,1" "2,1" "4,1" "5,1" "6,1" "2,2"}; do
  ATLAS_AMD_LEG_CFG=$c python bench.py --steps 6 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg $c legendre ms', round(d['roofline_kernels'][0]['avg_ms'],3), 'TF/s', round(d['roofline_kernels'][0]['achieved'],1))"
done
