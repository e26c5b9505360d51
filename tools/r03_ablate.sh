#!/bin/bash
# Dev tool (GPU box; ATLAS_AMD_LIB = -DAA_FFT_ABLATE build): one class, serial, ablation bits: 1 gather addresses collapsed,
# 2 pre+chirp, 4 filter, 8 chirp (store phase), 16 stores, 32 no gather at all
export ATLAS_AMD_LIB=atlas_amd/lib/dev/libatlas_amd_abl.so ATLAS_AMD_FFT_STREAMS=1
for M in 4096 5120; do for a in 0 1 32 38 63; do
  ATLAS_AMD_FFT_ONLY_M=$M ATLAS_AMD_FFT_ABLATE=$a python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('M=$M ablate=$a fourier ms', round(d['roofline_kernels'][1]['avg_ms'],3))"
done; done
