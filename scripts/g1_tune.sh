#!/bin/bash
# Tuning aid (run on the GPU box): rebuild the one-lane cartpole kernel with a register budget B (state beyond it goes
# to LDS) and LDS state layout R (TMPC_LDS_ROWS), relink, time the headline workload.  usage: scripts/g1_tune.sh "B:R ..."
cd "$(dirname "$0")/.." || exit 1
C=tinympc-julia_amd/csrc
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['config']['kernel'], 'kernel_ms=%.4f ms_per_step=%.4f' % (d['roofline']['kernel_ms'], d['ms_per_step']))"; }
cp tinympc-julia_amd/lib/libtinympc_hip.so /tmp/lib_orig.so
for br in ${1:-520:0}; do
  b=${br%%:*}; r=${br##*:}
  sed "s/TMPC_DEFINE_QUAD_ENTRY(4, 1, 20, 1, 520, 520, 3)/TMPC_DEFINE_QUAD_ENTRY(4, 1, 20, 1, $b, $b, 3)/" $C/inst_4_1_20_g1.hip > /tmp/g1x.hip
  /opt/rocm/bin/hipcc -I$C -DTMPC_LDS_ROWS=$r -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-honor-nans -c /tmp/g1x.hip -o /tmp/g1x.o || exit 1
  objs=$(for f in $C/*.hip $C/*.cpp; do n=$(basename ${f%.*}); [ $n != inst_4_1_20_g1 ] && echo $C/build/$n.o; done)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tinympc-julia_amd/lib/libtinympc_hip.so $objs /tmp/g1x.o || exit 1
  for p in 0 1; do
    timeout -k 10 300 python bench.py --precision $p --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | line "budget=$b lds_rows=$r precision=$p"
  done
done
cp /tmp/lib_orig.so tinympc-julia_amd/lib/libtinympc_hip.so
