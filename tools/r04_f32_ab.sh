#!/bin/bash
# Dev tool (GPU box): fp32 configurations under environment switches: tools/r04_f32_ab.sh "C5,C4f32" "ENV=.." "ENV=.." ...
export TMPDIR=/tmp
CFG=$1; shift
for rep in 1 2; do for e in "$@"; do
  env $e python tools/bench_configs.py --only $CFG --out /tmp/f32ab.jsonl 2>/dev/null | python3 -c "
import json,sys
for ln in sys.stdin.read().strip().splitlines():
    if not ln.startswith('{'): continue
    d=json.loads(ln); print('$e', d['metric'][-22:], round(d['ms_per_step'],3), [(round(k['avg_ms'],3)) for k in d['roofline_kernels']])"
done; done
