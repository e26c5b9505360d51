export ATLAS_AMD_FFT_STREAMS=1
ATLAS_AMD_FFT_DEBUG=1 python tools/fft_ab.py ATLAS_AMD_X 0 0 2>&1 | grep "fft ct" | sort -u
for M in 2048 2560 4096 5120; do
 for pad in 0 16384 40000 82000; do
  echo -n "M=$M pad=$pad: "; ATLAS_AMD_FFT_ONLY_M=$M python tools/fft_ab.py ATLAS_AMD_FFT_LDS_PAD $pad $pad 2>&1 | grep fourier | head -1
 done
done
