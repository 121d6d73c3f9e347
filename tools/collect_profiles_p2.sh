#!/bin/bash
# runs on the GPU box from the repo root: the round-5 evidence of the 'fp32_storage' mode on hi/lo planes (csrc/planes*.hip):
# kernel-trace stats + timelines (bs 8 and bs 1), SQ counters, HBM traffic counters -- each rocprofv3 --pmc in its own pass.
# usage: tools/collect_profiles_p2.sh r05      -> gpurun_out/prof_r05/
TAG=${1:-r05}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
for BS in 8 1; do
  cd /tmp; rm -rf /tmp/p_kp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kp -o kp -- python $R/tools/bench_precise.py 12 $BS > $OUT/${TAG}_precise_bs${BS}_bench_under_rocprof.json 2>/tmp/kp.err
  find /tmp/p_kp -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_precise_bs${BS}_kernel_stats.csv \;
  T=$(find /tmp/p_kp -name "*kernel_trace.csv" | head -1)
  python $R/tools/timing/timeline.py $T k_pl_stem | sed 's/(anonymous namespace):://g; s/pl:://g' > $OUT/${TAG}_precise_bs${BS}_timeline.txt
done
cd $R
tools/pmc_p2.sh $TAG 8 > /dev/null; cp gpurun_out/${TAG}_pmc_sq_counters.txt $OUT/${TAG}_precise_pmc_sq_counters.txt
cd /tmp; rm -rf /tmp/p_f /tmp/p_w
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_f -o f -- python $R/tools/bench_precise.py 3 8 > /tmp/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_w -o w -- python $R/tools/bench_precise.py 3 8 > /tmp/w.log 2>&1
F=$(find /tmp/p_f -name "*.db" | head -1); W=$(find /tmp/p_w -name "*.db" | head -1)
python $R/tools/pmc_traffic.py $F $W $OUT/${TAG}_precise_pmc_hbm_traffic_raw.json > /tmp/pmc.log 2>&1; tail -3 /tmp/pmc.log
ls -la $OUT
