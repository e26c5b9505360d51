#!/bin/bash
# Dev tool (GPU box): Legendre/Fourier software-pipeline sweep
for p in 1 2 3; do
  echo "== ATLAS_AMD_PIPELINE=$p"
  ATLAS_AMD_PIPELINE=$p python bench.py --steps 10 --warmup 3 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], [ (k['kernel'][:12], round(k['avg_ms'],2)) for k in d['roofline_kernels']])"
done
ATLAS_AMD_PIPELINE=3 python -m pytest tests/test_gpu_trans.py -m gpu -x -q 2>&1 | tail -2
