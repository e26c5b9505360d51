#!/bin/bash
# Dev tool (GPU box): the round's evidence in one call.
#   2. rocprofv3 --kernel-trace --stats of the bench command  -> gpurun_out/round/stats/
#   3. PMC passes (counters only): FETCH_SIZE, WRITE_SIZE, MFMA/LDS counters -> gpurun_out/round/pmc_*/
#   4. summary: per-kernel average duration, HBM traffic per launch -> gpurun_out/round/{kernel_stats.txt,pmc_traffic.json}
#   5. default bench (with cpu_baseline), the PMC profile of this run attached -> gpurun_out/round/bench_default.json
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/round
rm -rf $O; mkdir -p $O
cd /tmp
BENCH="python $R/bench.py --no-cpu-baseline"   # the default run (10 steps, 3 warm-up steps) without the CPU leg
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- $BENCH > $O/stats.log 2>&1
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" \
            "sq SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU"; do
  set -- $pass; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_$name --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_$name.log 2>&1
done
cd $R
python3 tools/prof_round_summary.py $O > $O/summary.txt 2>&1
tail -40 $O/summary.txt
# the default bench LAST, with this run's PMC profile in place (profiles/rNN_pmc_traffic.json of the round, stamped with the digest
# of the kernel sources): the line then carries `traffic` from the counters of the same sources on the same box
cp $O/pmc_traffic.json profiles/${PROF_ROUND_TAG:-r03}_pmc_traffic.json
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 400 $O/bench_default.json
