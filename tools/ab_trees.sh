#!/bin/bash
# Dev tool (GPU box): alternating runs of the headline bench from two source trees in ONE lease (same box, same thermal
# history): tells a slow box from a slow build.  Round 6: the round-4 tree (commit cc43b46, copied to _ab_r4/ and built
# there; not committed) against HEAD.
#   tools/ab_trees.sh [-r REPS] [-s STEPS] TREE_A TREE_B        (a tree = a directory holding bench.py + atlas_amd/)
export TMPDIR=/tmp
REPS=4; STEPS=20
while getopts "r:s:" o; do case $o in r) REPS=$OPTARG;; s) STEPS=$OPTARG;; esac; done
shift $((OPTIND - 1))
smi() { rocm-smi --showclocks --showpower --showtemp --json 2>/dev/null | python3 -c "
import json, sys
try:
    d = json.load(sys.stdin); c = d[sorted(d)[0]]
    keep = {k: v for k, v in c.items() if any(s in k.lower() for s in ('sclk', 'mclk', 'fclk', 'power', 'temperature (sensor junction)', 'temperature (sensor memory)'))}
    print('   smi', json.dumps(keep))
except Exception as e:
    print('   smi unavailable', e)"; }
for rep in $(seq $REPS); do for tree in "$@"; do
  smi
  ( cd "$tree" && python bench.py --steps $STEPS --warmup 3 --no-cpu-baseline 2>/dev/null ) | python3 -c "
import json, sys
for ln in sys.stdin.read().strip().splitlines():
    if not ln.startswith('{'):
        continue
    d = json.loads(ln)
    k = d.get('roofline_kernels', [])
    print('rep $rep [$tree]', 'ms/step', round(d['ms_per_step'], 4), 'legendre', round(k[0]['avg_ms'], 4), 'fourier', round(k[1]['avg_ms'], 4),
          'sustained_mfma_TF', round(k[0].get('peak_sustained_measured') or 0, 2), 'frac_of_sustained', round(k[0].get('frac_of_sustained') or 0, 4))"
done; done
smi
