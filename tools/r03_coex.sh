#!/bin/bash
# Dev tool (GPU box): co-residency probe (tools/coex_probe.py) with the product library and the -DAA_COEX build
#   make -C atlas_amd/csrc BUILD=build_coex LIBNAME=dev/libatlas_amd_coex.so HIPFLAGS_EXTRA=-DAA_COEX
export TMPDIR=/tmp
O=gpurun_out/r03coex; mkdir -p $O
V=atlas_amd/lib/dev/libatlas_amd_coex.so
python tools/coex_probe.py > $O/product.txt 2>&1
ATLAS_AMD_LIB=$V python tools/coex_probe.py > $O/coex.txt 2>&1
ATLAS_AMD_LIB=$V ATLAS_AMD_FFT_FAST_M=0 python tools/coex_probe.py > $O/coex_plain.txt 2>&1
ATLAS_AMD_FFT_FAST_M=0 python tools/coex_probe.py > $O/product_plain.txt 2>&1
grep -h "ms per transform" $O/product.txt $O/coex.txt $O/coex_plain.txt $O/product_plain.txt
grep -L "ms per transform" $O/*.txt | xargs -r tail -5
