#!/bin/bash
# Dev tool (GPU box): per-wavefront traces of two classes with the phase-end stamps (8 words) and with the stamps inside phase 0 (16)
R=$PWD; O=$R/gpurun_out/r03trace5; mkdir -p $O
for M in ${1:-4096 5120}; do
  for v in trace trace0; do
    W=8; [ $v = trace0 ] && W=16
    FFT_TRACE_WORDS=$W ATLAS_AMD_LIB=$R/atlas_amd/lib/dev/libatlas_amd_$v.so ATLAS_AMD_FFT_ONLY_M=$M ATLAS_AMD_FFT_STREAMS=1 FFT_TRACE_SAVE=$O/${v}_$M.npz \
      python tools/fft_trace.py > $O/${v}_$M.log 2>&1
    python tools/fft_trace_analyze.py $O/${v}_$M.npz 2>&1 | head -22
  done
done
