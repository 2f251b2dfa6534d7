#!/bin/bash
# kernel time vs batch for one config: scripts/batch_sweep.sh <config> "<batches>" [env...]
cfg=$1; shift; bs=$1; shift
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['config']['batch_per_gpu']; print('$cfg', d['config']['kernel'], 'batch', b, 'kernel_ms=%.3f us/inst=%.4f' % (d['roofline']['kernel_ms'], 1e3*d['roofline']['kernel_ms']/b))"; }
for b in $bs; do
  env "$@" timeout -k 10 300 python bench.py --config $cfg --batch $b --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | line
done
