#!/bin/bash
# Round profile of the bench command on configs 2, 3 and 4: rocprofv3 kernel-trace stats, then FETCH_SIZE and WRITE_SIZE
# in separate --pmc passes (the gfx950 guide's prescription), then SQ issue / wait counters in further passes.  Run through gpurun; outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp
# name:bench flags — the three configs cold, and the calling patterns of config 4 / the headline that move other bytes and issue other instructions
declare -A FLAGS=( [cartpole]="--config cartpole" [quadrotor]="--config quadrotor" [rocket_soc]="--config rocket_soc"
                   [rocket_soc_workspace_kept]="--config rocket_soc --keep-workspace"
                   [rocket_soc_check_live]="--config rocket_soc --tol 1e-30 --check-termination 1"
                   [cartpole_check_live]="--config cartpole --tol 1e-30 --check-termination 1" )
for cfg in ${CFGS:-cartpole quadrotor rocket_soc rocket_soc_workspace_kept rocket_soc_check_live cartpole_check_live}; do
  fl="${FLAGS[$cfg]} --no-cpu-baseline --no-extras"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_$cfg -- python3 $R/bench.py $fl --steps 10 --warmup 2 > $R/gpurun_out/prof_${tag}_$cfg.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcF_${tag}_$cfg -- python3 $R/bench.py $fl --steps 3 --warmup 1 > /dev/null 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcW_${tag}_$cfg -- python3 $R/bench.py $fl --steps 3 --warmup 1 > /dev/null 2>&1 || exit 1
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
    n=$(echo $grp | tr ' ' '_')
    timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmcS_${tag}_${cfg}_$n -- python3 $R/bench.py $fl --steps 3 --warmup 1 > /dev/null 2>&1 || exit 1
  done
  if [[ $cfg != cartpole* ]]; then  # matrix-core kernels: MFMA issue / busy counters (optional: skipped if the names are unknown)
    timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/pmcS_${tag}_${cfg}_MFMA -- python3 $R/bench.py $fl --steps 3 --warmup 1 > /dev/null 2>&1 || echo "MFMA counter pass skipped"
  fi
  echo "profiled $cfg"
done
cd $R && python bench.py --steps 20 --warmup 3 > gpurun_out/bench_${tag}_default.json 2> gpurun_out/bench_${tag}_default.err
tail -c 1500 gpurun_out/bench_${tag}_default.json
