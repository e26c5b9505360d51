#!/bin/bash
# Dev tool (GPU box; -DAA_FFT_ABLATE build): what the TABLE loads of the Fourier stage cost with the gather in place: ablation bits
# 2 pre + chirp of phase 0, 4 filter spectrum, 8 chirp of the store phase (addresses collapsed onto one line; results wrong, timing only);
# whole stage on the default four streams and the two biggest classes alone
export ATLAS_AMD_LIB=atlas_amd/lib/dev/libatlas_amd_abl.so
for rep in 1 2; do for a in 0 2 6 14; do
  ATLAS_AMD_FFT_ABLATE=$a python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stage ablate=$a fourier ms', round(d['roofline_kernels'][1]['avg_ms'],3))"
done; done
for M in 3840 5120; do for a in 0 2 14; do
  ATLAS_AMD_FFT_STREAMS=1 ATLAS_AMD_FFT_ONLY_M=$M ATLAS_AMD_FFT_ABLATE=$a python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('M=$M ablate=$a fourier ms', round(d['roofline_kernels'][1]['avg_ms'],3))"
done; done
