#!/bin/bash
# one config under the three loop variants: scripts/gpu_modes.sh <config> [group]
cfg=${1:-cartpole}; grp=${2:-}
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['config']['kernel'], d['dtype'], 'kernel_ms=%.3f' % d['roofline']['kernel_ms'])"; }
for p in 0 1; do
  for mode in "UNI" "OS:TINYMPC_HIP_NO_UNI=1" "BASE:TINYMPC_HIP_NO_UNI=1 TINYMPC_HIP_NO_OS=1"; do
    name=${mode%%:*}; envs=${mode#*:}; [ "$envs" = "$mode" ] && envs=""
    env $envs ${grp:+TINYMPC_HIP_GROUP=$grp} python bench.py --config $cfg --precision $p --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | line "$cfg/$name"
  done; done
