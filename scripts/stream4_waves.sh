#!/bin/bash
# Tuning aid (run on the GPU box): rebuild the 4-lane stream kernels held to W wavefronts per SIMD and time
# config 4 (rocket, cones + affine term) and a fallback shape.  usage: scripts/stream4_waves.sh "2 3 4"
cd "$(dirname "$0")/.." || exit 1
C=tinympc-julia_amd/csrc
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['config']['kernel'], d['dtype'], 'solves/s=%.3e kernel_ms=%.3f' % (d['value'], d['roofline']['kernel_ms']))"; }
for w in ${1:-2 3 4}; do
  mkdir -p $C/build_w$w
  for f in $C/sinst_q_*.hip; do
    /opt/rocm/bin/hipcc -DTMPC_STREAM4_WAVES=$w -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-honor-nans -c $f -o $C/build_w$w/$(basename ${f%.hip}).o &
  done; wait
  objs=$(ls $C/build/*.o | grep -v sinst_q_)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tinympc-julia_amd/lib/libtinympc_hip.so $objs $C/build_w$w/*.o || exit 1
  for p in 0 1; do
    timeout -k 10 300 python bench.py --config rocket_soc --precision $p --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | line "waves=$w rocket_soc"
    TINYMPC_HIP_NO_QUAD=1 timeout -k 10 300 python bench.py --config rocket --precision $p --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | line "waves=$w rocket(stream)"
    TINYMPC_HIP_NO_QUAD=1 timeout -k 10 300 python bench.py --config quadrotor --precision $p --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | line "waves=$w quadrotor(stream)"
  done
done
