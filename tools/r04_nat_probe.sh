#!/bin/bash
# Dev tool (GPU box): the native mixed-radix rows alone under occupancy pads and with the gather left out (dev build -DAA_FFT_ABLATE)
for pad in 0 14000 42000; do echo "== LDS pad $pad"; ATLAS_AMD_FFT_DEBUG=1 ATLAS_AMD_FFT_LDS_PAD=$pad python tools/fft_native_prof.py 2>&1 | grep -E "fourier ms|workgroups/CU" | sort -u; done
echo "== ablate build, abl=0 / 32 (no gather)"
for a in 0 32; do ATLAS_AMD_LIB=$PWD/atlas_amd/lib/dev/libatlas_amd_abl.so ATLAS_AMD_FFT_ABLATE=$a python tools/fft_native_prof.py 2>&1 | grep -E "fourier ms"; done
