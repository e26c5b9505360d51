#!/bin/bash
# Dev tool (GPU box): alternating A/B runs of the bench under (library build, environment) variants -- the one script behind the
# round-3 / round-4 experiment logs (it replaces tools/r03_ab3 / ab4 / ab_lib / ab_lib2 / env_sweep / prefetch / streams / fastm /
# leg_ab / c2 / c5 / leg_f32_sweep / ablate / ablate2 and r04_f32_ab).
#   tools/ab.sh [-r REPS] [-c CFG[,CFG]] [-s STEPS] VARIANT [VARIANT ...]
#   VARIANT = LIB[:ENV=..[,ENV=..]]   LIB = product | <name>  (atlas_amd/lib/dev/libatlas_amd_<name>.so, e.g. abl, exp, trace)
#   -c: a configuration of tools/bench_configs.py (C2, C3x1, C5, C5n, C4f32, ...) instead of the headline bench.py
# Examples (what the logs under profiles/ were taken with):
#   tools/ab.sh product product:ATLAS_AMD_FFT_NATIVE=1                       native mixed-radix rows on / off
#   tools/ab.sh product:ATLAS_AMD_FFT_STREAMS=1 product:ATLAS_AMD_FFT_STREAMS=4 product:ATLAS_AMD_FFT_STREAMS=6
#   tools/ab.sh product:ATLAS_AMD_FFT_PREFETCH=0 product:ATLAS_AMD_FFT_PREFETCH=2 product:ATLAS_AMD_FFT_PREFETCH=8,2
#   tools/ab.sh product:ATLAS_AMD_LEG_KERNEL=lean product:ATLAS_AMD_LEG_KERNEL=classic exp:ATLAS_AMD_LEG_KERNEL=lean2
#   tools/ab.sh abl:ATLAS_AMD_FFT_ABLATE=0 abl:ATLAS_AMD_FFT_ABLATE=64 abl:ATLAS_AMD_FFT_ABLATE=1      (results wrong by construction)
#   tools/ab.sh abl:ATLAS_AMD_FFT_ONLY_M=5120,ATLAS_AMD_FFT_STREAMS=1,ATLAS_AMD_FFT_ABLATE=32 ...        one row class
#   tools/ab.sh -c C2 product:ATLAS_AMD_FFT_COARSE_FUSED=1 product:ATLAS_AMD_FFT_COARSE_FUSED=0
#   tools/ab.sh -c C5,C4f32 product:ATLAS_AMD_FFT_GROUP_LOG2=3 product:ATLAS_AMD_FFT_GROUP_LOG2=4
#   tools/ab.sh -c C5,C4f32,C5n product:ATLAS_AMD_FFT_F32_PAIRS=0 product:ATLAS_AMD_FFT_F32_PAIRS=1                 fp32 rows: one / two fields per job
#   tools/ab.sh -c C5 product:ATLAS_AMD_LEG_CFG=3,2 product:ATLAS_AMD_LEG_CFG=5,2                        (a comma inside a value: use ';')
export TMPDIR=/tmp
REPS=2; CFG=""; STEPS=10
while getopts "r:c:s:" o; do case $o in r) REPS=$OPTARG;; c) CFG=$OPTARG;; s) STEPS=$OPTARG;; esac; done
shift $((OPTIND - 1))
for rep in $(seq $REPS); do for v in "$@"; do
  lib=${v%%:*}; envs=""; [[ "$v" == *:* ]] && envs=${v#*:}
  # ENV separators: ',' between assignments (an assignment starts with an upper-case name followed by '='); ';' always separates
  envs=$(echo "$envs" | sed -E 's/,([A-Z_][A-Z0-9_]*=)/ \1/g; s/;/ /g')
  if [ "$lib" = product ]; then unset ATLAS_AMD_LIB; else export ATLAS_AMD_LIB=$PWD/atlas_amd/lib/dev/libatlas_amd_$lib.so; fi
  if [ -z "$CFG" ]; then
    env $envs python bench.py --steps $STEPS --warmup 3 --no-cpu-baseline 2>/dev/null
  else
    env $envs python tools/bench_configs.py --only $CFG --out /tmp/ab_cfg.jsonl 2>/dev/null
  fi | python3 -c "
import json, sys
for ln in sys.stdin.read().strip().splitlines():
    if not ln.startswith('{'):
        continue
    d = json.loads(ln)
    print('rep $rep [$v]', d['metric'][-24:], 'ms/step', round(d['ms_per_step'], 4), [(k['kernel'][:10], round(k['avg_ms'], 4)) for k in d.get('roofline_kernels', [])])"
done; done
