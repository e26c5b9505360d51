#!/bin/bash
# First GPU call of the next round: everything round 1 built after its GPU budget was spent, in one run.
#   gpurun --timeout 900 -- 'bash tools/round2_first_call.sh'
# Outputs under gpurun_out/r02_first/.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02_first
mkdir -p "$OUT"
export TMPDIR=/tmp

echo "== 2. headline bench, single GPU" | tee -a "$OUT/summary.txt"
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-baseline-blas 2>&1 | tail -1 > "$OUT/bench_n1.json"
python - << 'PY' | tee -a "$OUT/summary.txt"
import json
d = json.load(open("gpurun_out/r02_first/bench_n1.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"])
print("cpu_baseline", d.get("cpu_baseline", {}).get("value"), "blas", d.get("cpu_baseline_blas", {}).get("value"))
PY

echo "== 3. Legendre table set-up at T1279: host + upload vs device generation" | tee -a "$OUT/summary.txt"
timeout 300 python - << 'PY' 2>&1 | tee -a "$OUT/summary.txt"
import time, numpy as np, atlas_amd
g = atlas_amd.Grid("O1280")
for tables in ("host", "device", "device"):
    t0 = time.perf_counter()
    tr = atlas_amd.Trans(g, 1279, tables=tables)
    tr.synchronize()
    print(f"tables={tables}: Trans set-up {time.perf_counter() - t0:.2f} s", flush=True)
    del tr
PY

echo "== 5. mirror-band parity at full size (child process of the GPU test, TL1279 would need the oracle: O640 here)" | tee -a "$OUT/summary.txt"
timeout 300 python tests/mirror_check.py O640 639 20 4 2>&1 | tail -2 | tee -a "$OUT/summary.txt"
