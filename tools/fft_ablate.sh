#!/bin/bash
# Dev tool (GPU box; library built with HIPFLAGS_EXTRA=-DAA_FFT_ABLATE): time one FFT class with classes of global
# accesses collapsed onto one cache line. bits: 1 gather, 2 pre+chirp (load phase), 4 filter, 8 chirp (store phase), 16 stores
M=${1:-5120}
for a in 0 1 2 4 8 16 3 31; do
  ATLAS_AMD_FFT_ONLY_M=$M ATLAS_AMD_FFT_ABLATE=$a python bench.py --steps 5 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('M=$M ablate=$a fourier ms', round(d['roofline_kernels'][1]['avg_ms'],3))"
done
