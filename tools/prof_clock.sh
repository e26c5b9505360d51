#!/bin/bash
# Dev tool (GPU box): effective shader clock of each kernel = GRBM_GUI_ACTIVE / kernel duration
export TMPDIR=/tmp
R=$PWD
rm -rf $R/gpurun_out/clk; mkdir -p $R/gpurun_out/clk
cd /tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $R/gpurun_out/clk --output-format csv -- python $R/bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline > $R/gpurun_out/clk/bench.log 2>&1
cd $R
python3 - << 'PY'
import csv, glob, collections
cc = glob.glob('gpurun_out/clk/**/*counter_collection.csv', recursive=True)[0]
kt = glob.glob('gpurun_out/clk/**/*kernel_trace.csv', recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'])
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cc)):
    if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
    d, name = dur[r['Dispatch_Id']]
    a = agg[name[:70]]
    a[0] += float(r['Counter_Value']); a[1] += d; a[2] += 1
for k, (cyc, ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
    print("%-72s n=%3d  avg %.3f ms  GUI_ACTIVE/ns = %.3f (GHz if summed over 1 counter instance)" % (k, n, ns / n / 1e6, cyc / ns))
PY
