#!/bin/bash
# Round 5: A/B of the current library against the round-4 one on configs[2] + the level-set variants of the tile kernel.  gpurun_out/$R5_OUT
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${R5_OUT:-r05c}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tile or tens or config3 or golden or covs or notebook or lane" > $O/pytest_tile.log 2>&1; echo "rc=$?" >> $O/pytest_tile.log
tail -3 $O/pytest_tile.log
for rnd in 1 2 3; do
  for cfg in "c3" "c3 --increments"; do
    for lib in default libgpsig_hip_r4.so ${AB_LIBS:-}; do
      if [ $lib = default ]; then unset GPSIG_LIB; else export GPSIG_LIB=$PWD/gpsig_amd/lib/$lib; fi
      timeout 300 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $rnd  $cfg  lib=$lib  kernel ms %.3f  ms/step %.3f  rel_err %.2e  clock %.2f' % (d['roofline']['kernel_ms_per_launch'], d['ms_per_step'], d['rel_err'], d['clock_ghz']))" >> $O/ab_c3.txt 2>&1
    done
  done
done
unset GPSIG_LIB
cat $O/ab_c3.txt
timeout 600 python tools/bench_c3.py > $O/bench_c3_variants.txt 2>&1
grep -v "tensor lanes" $O/bench_c3_variants.txt | cut -c1-150
