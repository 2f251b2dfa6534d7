#!/bin/bash
# Round profile of the bench command on configs 2, 3 and 4: rocprofv3 kernel-trace stats, then FETCH_SIZE and WRITE_SIZE
# in separate --pmc passes (the gfx950 guide's prescription), then SQ issue / wait counters in further passes.  Run through gpurun; outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp
for cfg in ${CFGS:-cartpole quadrotor rocket_soc}; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_$cfg -- python3 $R/bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_${tag}_$cfg.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcF_${tag}_$cfg -- python3 $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcW_${tag}_$cfg -- python3 $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 1
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
    n=$(echo $grp | tr ' ' '_')
    timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmcS_${tag}_${cfg}_$n -- python3 $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 1
  done
  if [ $cfg != cartpole ]; then  # matrix-core kernels: MFMA issue / busy counters (optional: skipped if the names are unknown)
    timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/pmcS_${tag}_${cfg}_MFMA -- python3 $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || echo "MFMA counter pass skipped"
  fi
done
cd $R && python bench.py --steps 20 --warmup 3 > gpurun_out/bench_${tag}_default.json 2> gpurun_out/bench_${tag}_default.err
tail -c 1500 gpurun_out/bench_${tag}_default.json
