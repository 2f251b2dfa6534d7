#!/bin/bash
# SQ counter passes of one bench config (rocprofv3 --pmc, kernel-trace off): scripts/pmc_one.sh <config> <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
cfg=${1:-rocket_soc}
tag=${2:-x}
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
  n=$(echo $grp | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmcS_${tag}_${cfg}_$n -- python3 $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || echo "pass $n failed"
done
python3 - <<PY
import csv, glob
acc = {}
for d in sorted(glob.glob("$R/gpurun_out/pmcS_${tag}_${cfg}_*")):
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "admm" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
