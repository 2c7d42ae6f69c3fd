#!/bin/bash
# A/B of library builds (GPSIG_LIB), alternating processes on one box:  AB_CFGS="c2 --base rbf|c3" tools/gpu_ab_libs.sh libA.so libB.so ...
cd "$(dirname "$0")/.."
IFS='|' read -ra CFGS <<< "${AB_CFGS:-c2 --base rbf}"
for rnd in 1 2 3; do
  for cfg in "${CFGS[@]}"; do
    for lib in default "$@"; do
      if [ $lib = default ]; then unset GPSIG_LIB; else export GPSIG_LIB=$PWD/gpsig_amd/lib/$lib; fi
      timeout 300 python bench.py --config $cfg --steps ${AB_STEPS:-10} --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $rnd  $cfg  lib=$lib  kernel ms %.3f  ms/step %.3f  rel_err %.2e  clock %.2f' % (d['roofline']['kernel_ms_per_launch'], d['ms_per_step'], d['rel_err'], d['clock_ghz']))"
    done
  done
done
