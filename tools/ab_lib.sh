#!/bin/bash
# A/B of two builds of the library on one box: the default one against gpsig_amd/lib/$1 (GPSIG_LIB), alternating processes.
# usage: tools/ab_lib.sh libgpsig_hip_lds.so "c2" "c2 --base rbf" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
alt=$1; shift
for rnd in 1 2 3; do
  for cfg in "$@"; do
    for lib in default $alt; do
      if [ $lib = default ]; then unset GPSIG_LIB; else export GPSIG_LIB=$PWD/gpsig_amd/lib/$lib; fi
      python bench.py --config $cfg --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $rnd  $cfg  lib=$lib  kernel ms %.3f  ms/step %.3f  rel_err %.2e' % (d['roofline']['kernel_ms_per_launch'], d['ms_per_step'], d['rel_err']))"
    done
  done
done
