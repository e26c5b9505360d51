#!/bin/bash
# Dev tool (GPU box): the fp32 configurations (C5 = F1280, C5n = N1280) and the headline, stage times
for c in ${1:-C5 C5n}; do python tools/bench_configs.py --only $c 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c', round(d['value'],2), round(d['ms_per_step'],3), [(round(k['avg_ms'],3),round(k['frac'],3)) for k in d['roofline_kernels']])"; done
