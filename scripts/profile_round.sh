#!/bin/bash
# Round profile of the default bench command: rocprofv3 kernel-trace stats, then FETCH_SIZE and WRITE_SIZE
# in separate --pmc passes (the gfx950 guide's prescription).  Run through gpurun; outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp
for cfg in cartpole quadrotor; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_$cfg -- python3 $R/bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_${tag}_$cfg.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcF_${tag}_$cfg -- python3 $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcW_${tag}_$cfg -- python3 $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 1
done
cd $R && python bench.py --steps 20 --warmup 3 > gpurun_out/bench_${tag}_default.json 2> gpurun_out/bench_${tag}_default.err
tail -c 1500 gpurun_out/bench_${tag}_default.json
