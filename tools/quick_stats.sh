#!/bin/bash
# Dev tool (GPU box): per-kernel durations of the bench (rocprofv3 kernel trace) -> gpurun_out/quick/
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/quick
rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace -d $O/stats --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
cd $R
tail -1 $O/stats.log | cut -c1-400
python3 tools/prof_round_summary.py $O 2>&1 | head -28
