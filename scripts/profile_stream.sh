#!/bin/bash
# rocprofv3 kernel trace + PMC passes of the stream kernel on config 4 (run on the GPU box)
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_stream_$tag; mkdir -p $O
cmd="python3 $R/bench.py --config rocket_soc --steps 5 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- $cmd > $O/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
  n=$(echo $c | tr ' ' '_'); rocprofv3 --pmc $c -d $O/pmc_$n -o p -- $cmd > $O/pmc_$n.log 2>&1
done
find $O -name "*.csv" | head -40
