#!/bin/bash
# usage: scripts/gpu_quick.sh <tag> [notest]  — gpu tests + short bench lines for every config
tag=${1:-run}
if [ "$2" != "notest" ]; then timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$tag.log 2>&1; tail -3 gpurun_out/pytest_gpu_$tag.log; fi
rm -f gpurun_out/bench_$tag.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['config']['kernel'], d['dtype'], 'solves/s=%.3e kernel_ms=%.3f valu=%.3f' % (d['value'], d['roofline']['kernel_ms'], d['valu']['frac']))"; }
for g in 4 2 1; do for p in 0 1; do
  TINYMPC_HIP_GROUP=$g timeout -k 10 200 python bench.py --config cartpole --precision $p --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | line cartpole >> gpurun_out/bench_$tag.log
done; done
for c in "quadrotor 0" "quadrotor 1" "rocket 0" "rocket 1"; do set -- $c
  timeout -k 10 200 python bench.py --config $1 --precision $2 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | line $1 >> gpurun_out/bench_$tag.log
done
cat gpurun_out/bench_$tag.log
