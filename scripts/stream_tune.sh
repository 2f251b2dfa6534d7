#!/bin/bash
# Tuning aid (run on the GPU box): rebuild the stream kernels of the (6, x) shapes with W wavefronts per SIMD and
# prefetch depth D, and time config 4 and the box-only rocket on them.  usage: scripts/stream_tune.sh "W:D W:D ..."
cd "$(dirname "$0")/.." || exit 1
C=tinympc-julia_amd/csrc
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['config']['kernel'], 'batch', d['config']['batch_per_gpu'], 'kernel_ms=%.3f' % d['roofline']['kernel_ms'])"; }
cp tinympc-julia_amd/lib/libtinympc_hip.so /tmp/lib_orig.so
for wd in ${1:-3:1}; do
  w=${wd%%:*}; d=${wd##*:}
  /opt/rocm/bin/hipcc "-DTMPC_STREAM_WAVES(G)=$w" "-DTMPC_STREAM_DEPTH(G)=$d" -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-honor-nans -c $C/sinst_g4_3.hip -o /tmp/sinst_g4_3_$w$d.o || exit 1
  objs=$(for f in $C/*.hip $C/*.cpp; do n=$(basename ${f%.*}); [ $n != sinst_g4_3 ] && echo $C/build/$n.o; done)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tinympc-julia_amd/lib/libtinympc_hip.so $objs /tmp/sinst_g4_3_$w$d.o || exit 1
  for b in 8192 32768 65536; do
    timeout -k 10 300 python bench.py --config rocket_soc --batch $b --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | line "waves=$w depth=$d rocket_soc"
  done
  TINYMPC_HIP_NO_QUAD=1 timeout -k 10 300 python bench.py --config rocket --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | line "waves=$w depth=$d rocket(stream)"
done
cp /tmp/lib_orig.so tinympc-julia_amd/lib/libtinympc_hip.so
