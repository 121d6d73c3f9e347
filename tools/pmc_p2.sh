#!/bin/bash
# runs on the GPU box from the repo root: SQ counters (MFMA pipe busy, wait buckets, LDS conflicts) and the effective clock of
# the 'fp32_storage' kernels, per kernel name (rocprofv3 --pmc, two passes).  usage: tools/pmc_p2.sh <tag>
TAG=${1:-p2}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; rm -rf /tmp/p_sq1 /tmp/p_sq2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_sq1 -o s1 -- python $R/tools/bench_precise.py 3 ${2:-8} > /tmp/s1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES --output-format csv -d /tmp/p_sq2 -o s2 -- python $R/tools/bench_precise.py 3 ${2:-8} > /tmp/s2.log 2>&1
python - <<PY > $OUT/${TAG}_pmc_sq_counters.txt
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(list)
for d in ('/tmp/p_sq1', '/tmp/p_sq2'):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r'\(anonymous namespace\)::|pl::', '', r['Kernel_Name'])[:64] + ' g' + r.get('Grid_Size', '?')
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
            if r.get('Start_Timestamp') and r.get('End_Timestamp') and r['Counter_Name'] in ('SQ_WAVE_CYCLES', 'SQ_BUSY_CU_CYCLES'):
                dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print('# per-launch averages, rocprofv3 --pmc (two passes), tools/bench_precise.py; SQ_*_CYCLES count quad-cycles per wave; clock = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / duration, launches >= 20 us only')
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0)):
    row = {c: x / cnt[(k, c)] for c, x in v.items()}
    util = row.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * row['SQ_BUSY_CU_CYCLES']) if row.get('SQ_BUSY_CU_CYCLES') else float('nan')
    wc = row.get('SQ_WAVE_CYCLES', 0) or float('nan')
    du = sum(dur[k]) / max(len(dur[k]), 1)
    # (GRBM_GUI_ACTIVE / duration is only a clock for launches long against the counter's start / stop skew: below 20 us it read
    #  4.8-10 GHz in round 5 -- not printed there)
    clk = ('clock %.2f GHz' % (row.get('GRBM_GUI_ACTIVE', 0) / 8 / (du * 1e3))) if du >= 20 else 'clock   n/a   '
    print('%-80s n %3d  %7.1f us  %s  mfma_util %.3f  wait_any %.2f  wait_inst %.2f  active %.2f  lds_conf %.2f' % (
        k, cnt[(k, 'SQ_WAVE_CYCLES')], du, clk, util, row.get('SQ_WAIT_ANY', 0) / wc, row.get('SQ_WAIT_INST_ANY', 0) / wc,
        row.get('SQ_ACTIVE_INST_ANY', 0) / wc, row.get('SQ_LDS_BANK_CONFLICT', 0) / max(row.get('SQ_LDS_IDX_ACTIVE', 0), 1)))
    print('    ' + ' '.join('%s=%d' % (c, x) for c, x in sorted(row.items())))
PY
head -c 6000 $OUT/${TAG}_pmc_sq_counters.txt | grep -v "^    " | head -40
