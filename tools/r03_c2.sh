#!/bin/bash
# Dev tool (GPU box): C2 (TL159 -> O160, 60 fields) under environment switches: tools/r03_c2.sh "ENV=.. ENV=.." "ENV=.." ...
export TMPDIR=/tmp
for rep in 1 2; do for e in "$@"; do
  env $e python tools/bench_configs.py --only C2 --out /tmp/c2.jsonl 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', round(d['value'],1), round(d['ms_per_step'],4), [(round(k['avg_ms'],4)) for k in d['roofline_kernels']])"
done; done
