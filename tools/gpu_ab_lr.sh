#!/bin/bash
# A/B of builds of the fused low-rank feature kernel on one box (GPSIG_LIB), alternating processes
cd "$(dirname "$0")/.."
for rnd in 1 2; do
  for lib in default "$@"; do
    if [ $lib = default ]; then unset GPSIG_LIB; else export GPSIG_LIB=$PWD/gpsig_amd/lib/$lib; fi
    python tools/bench_lr.py --config c3 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $rnd lib=$lib  ms %.3f  seq features %.3f' % (d['ms_per_evaluation'], d['stages']['seq_features_ms']))"
  done
done
