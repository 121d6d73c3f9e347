#!/bin/bash
# round-2 GPU session A: new kernels + parity tests first (each under its own timeout), then the bench, then the full suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== block kernel tests"; timeout 300 python -m pytest tests/test_gpu_block.py -q -x --timeout 120 2>&1 | tail -15 | tee gpurun_out/a_block.log
echo "== ab_block"; timeout 200 python tools/ab_block.py 2>&1 | tail -12 | tee gpurun_out/a_ab_block.log
echo "== parity fullsize"; timeout 900 python -m pytest tests/test_gpu_parity_fullsize.py -q --timeout 600 -s 2>&1 | tail -80 | tee gpurun_out/a_parity.log
echo "== dist + config5"; timeout 600 python -m pytest tests/test_gpu_dist.py "tests/test_gpu_train.py::test_config5_widerface_s_640_loss_curve_vs_fp32_autograd" -q --timeout 400 -s 2>&1 | tail -40 | tee gpurun_out/a_dist_train.log
echo "== bench"; timeout 600 python bench.py --steps 50 --warmup 10 > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err; tail -c 3000 gpurun_out/a_bench.json; tail -5 gpurun_out/a_bench.err
echo "== bench (no block fusion)"; LFD_FUSED_BLOCK=0 timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-latency --no-train > gpurun_out/a_bench_nofuse.json 2>/dev/null; python - <<'PY'
import json
for f in ('a_bench', 'a_bench_nofuse'):
    try:
        d = json.load(open('gpurun_out/%s.json' % f)); print(f, d['ms_per_step'], d['value'], d.get('step_ms_hip_events'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
echo "== full gpu suite"; timeout 900 python -m pytest tests -q -m gpu --timeout 600 2>&1 | tail -15 | tee gpurun_out/a_full.log
