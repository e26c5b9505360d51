#!/bin/bash
# Dev tool (GPU box): role-split Legendre kernel (ATLAS_AMD_LEG_KERNEL) with parts left out (ATLAS_AMD_LEG_ABLATE bit mask:
# 2 table DMA, 4 spectra DMA, 32 MFMA steps, 64 every workgroup reads the first item's table block, 128 every workgroup
# reads the same four spectra rows).  Results are wrong with any bit set.
export ATLAS_AMD_LEG_KERNEL=${1:-split}
for a in $2; do
  python tools/fft_ab.py ATLAS_AMD_LEG_ABLATE $a $a 2>&1 | grep legendre | head -1
done
