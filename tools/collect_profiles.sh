#!/bin/bash
# runs on the GPU box from the repo root: kernel-trace stats, HBM traffic counters and SQ counters -- each in its OWN pass
# (rocprofv3 --pmc only together with --kernel-trace; FETCH_SIZE and WRITE_SIZE do not fit one pass).  usage: collect_profiles.sh r02
TAG=${1:-r03}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp
rm -rf /tmp/p_kt /tmp/p_f /tmp/p_w /tmp/p_sq1 /tmp/p_sq2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kt -o kt -- python $R/bench.py --steps 30 --warmup 5 --pipeline 1 --no-cpu-baseline --no-latency --no-train --no-siblings --no-configs --headline-mode fp16 > $OUT/${TAG}_bench_under_rocprof.json 2>/tmp/kt.err
find /tmp/p_kt -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
# the fp32-storage precision mode on the same workload (its own kernel set: k_p32_conv<...>, k_p32_gn_*)
rm -rf /tmp/p_kp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kp -o kp -- python $R/tools/bench_precise.py 10 > $OUT/${TAG}_precise_bench_under_rocprof.json 2>/tmp/kp.err
find /tmp/p_kp -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_precise_kernel_stats.csv \;
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_f -o f -- python $R/bench.py --steps 4 --warmup 1 --no-graph --no-cpu-baseline --no-latency --no-train --no-siblings --no-configs --headline-mode fp16 > /tmp/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_w -o w -- python $R/bench.py --steps 4 --warmup 1 --no-graph --no-cpu-baseline --no-latency --no-train --no-siblings --no-configs --headline-mode fp16 > /tmp/w.log 2>&1
F=$(find /tmp/p_f -name "*.db" | head -1); W=$(find /tmp/p_w -name "*.db" | head -1)
python $R/tools/pmc_traffic.py $F $W $OUT/${TAG}_pmc_hbm_traffic_raw.json > /tmp/pmc.log 2>&1; tail -3 /tmp/pmc.log
# SQ counters of the CURRENT kernels (MFMA busy, wait buckets, LDS conflicts): two passes of 8
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d /tmp/p_sq1 -o s1 -- python $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-latency --no-train --no-siblings --no-configs --headline-mode fp16 > /tmp/s1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES --output-format csv -d /tmp/p_sq2 -o s2 -- python $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-latency --no-train --no-siblings --no-configs --headline-mode fp16 > /tmp/s2.log 2>&1
python - <<PY > $OUT/${TAG}_pmc_sq_counters.txt
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for d in ('/tmp/p_sq1', '/tmp/p_sq2'):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])[:70]
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
print('# per-launch averages, rocprofv3 --pmc (two passes), bench.py --no-graph; SQ_*_CYCLES are in units of 4 clocks per wave/SIMD as the SQ counts them')
print('# MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)')
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0)):
    row = {c: x / cnt[(k, c)] for c, x in v.items()}
    util = row.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * row['SQ_BUSY_CU_CYCLES']) if row.get('SQ_BUSY_CU_CYCLES') else float('nan')
    wc = row.get('SQ_WAVE_CYCLES', 0) or float('nan')
    print('%-70s launches %3d  mfma_util %.3f  wait_any %.2f  wait_inst %.2f  active %.2f  lds_conflict/lds_active %.2f' % (
        k, cnt[(k, 'SQ_WAVE_CYCLES')], util, row.get('SQ_WAIT_ANY', 0) / wc, row.get('SQ_WAIT_INST_ANY', 0) / wc,
        row.get('SQ_ACTIVE_INST_ANY', 0) / wc, row.get('SQ_LDS_BANK_CONFLICT', 0) / max(row.get('SQ_LDS_IDX_ACTIVE', 0), 1)))
    print('    ' + ' '.join('%s=%d' % (c, x) for c, x in sorted(row.items())))
PY
ls -la $OUT; head -30 $OUT/${TAG}_pmc_sq_counters.txt
