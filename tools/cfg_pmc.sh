#!/bin/bash
# Dev tool (GPU box) [r4]: SQ counters of one configuration of tools/bench_configs.py (counters only; rocprofv3 --pmc with --kernel-trace),
# per kernel and dispatch: MFMA busy share, what the wavefronts wait for, instruction counts.
#   tools/cfg_pmc.sh C5 [outdir-tag]
export TMPDIR=/tmp
CFG=${1:-C5}
R=$PWD
O=$R/gpurun_out/${2:-cfgpmc}_$CFG
rm -rf $O; mkdir -p $O
cd /tmp
i=0
for pass in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
  "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $pass -d $O/p$i --output-format csv -- python $R/tools/bench_configs.py --only $CFG > $O/p$i.log 2>&1
done
cd $R
python3 - "$O" << 'PY'
import csv, glob, collections, sys
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob(O + '/p*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'fft_rows' not in k and 'legendre_kernel' not in k and 'legendre_lean' not in k: continue
        k = k.split('(')[0][-64:]
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); disp[(k, row['Counter_Name'])].add(row['Dispatch_Id'])
for k in sorted(agg, key=lambda k: -agg[k].get('SQ_WAVE_CYCLES', 0))[:12]:
    c = {n: v / len(disp[(k, n)]) for n, v in agg[k].items()}
    print(k)
    for n in sorted(c):
        print('    %-30s %.5g' % (n, c[n]))
    wc = c.get('SQ_WAVE_CYCLES', 0)
    if wc:
        print('    -> of wave time: waiting %.1f %%, wait-for-issue %.1f %%, active %.1f %% (VALU incl. MFMA %.1f %%, LDS %.1f %%)' % (
            100 * c.get('SQ_WAIT_ANY', 0) / wc, 100 * c.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc,
            100 * c.get('SQ_ACTIVE_INST_VALU', 0) / wc, 100 * c.get('SQ_ACTIVE_INST_LDS', 0) / wc))
    if c.get('SQ_BUSY_CYCLES') and c.get('SQ_VALU_MFMA_BUSY_CYCLES'):
        print('    -> MFMA busy / SQ busy cycles %.3f   MFMA instr %.4g  other VALU %.4g  LDS %.4g  SALU %.4g' % (
            c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES'], c.get('SQ_INSTS_MFMA', 0), c.get('SQ_INSTS_VALU', 0) - c.get('SQ_INSTS_MFMA', 0),
            c.get('SQ_INSTS_LDS', 0), c.get('SQ_INSTS_SALU', 0)))
PY
