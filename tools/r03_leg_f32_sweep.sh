#!/bin/bash
# Dev tool (GPU box): Legendre tiling sweep for the fp32 variant (C5: TL1279 -> F1280, 137 levels), ATLAS_AMD_LEG_CFG="rtw,nrg"
for c in ${@:-"3,2" "5,2" "9,1" "6,1" "9,2" "4,3" "3,3" "5,1"}; do
  echo -n "cfg $c: "; ATLAS_AMD_LEG_CFG=$c bash tools/r03_c5.sh C5
done
