#!/bin/bash
# A/B on one box: Kzx tile kernel with the 256-entry exp table / degree-4 tail (default build) against 1024 and 2048 entries / degree 3
# (gpsig_amd/lib/libgpsig_hip_e1024.so, _e2048.so: make LIBNAME=... EXTRA=-DTVS_EXPTAB=1024 OBJDIR=...), alternating processes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rnd in 1 2 3; do
  for cfg in "c3" "c3 --increments"; do
    for lib in default libgpsig_hip_e1024.so libgpsig_hip_e2048.so; do
      if [ $lib = default ]; then unset GPSIG_LIB; else export GPSIG_LIB=$PWD/gpsig_amd/lib/$lib; fi
      python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $rnd  $cfg  lib=$lib  Kzx kernel ms %.3f  ms/step %.3f  rel_err %.2e' % (d['roofline']['kernel_ms_per_launch'], d['ms_per_step'], d['rel_err']))"
    done
  done
done
