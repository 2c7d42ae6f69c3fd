#!/bin/bash
# A/B on one box: equal depth pieces (round 3) against the graded tail (round 4) of the feature contraction, alternating processes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04; mkdir -p $O
for rnd in 1 2 3; do
  for g in 0 1; do
    GPSIG_OPTIONS="sig_graded=$g" python bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --timed-loop-only 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $rnd  c2  sig_graded=$g  kernel ms %.3f  ms/step %.3f  rel_err %s  clock %s' % (d['roofline']['kernel_ms_per_launch'], d['ms_per_step'], d.get('rel_err'), d.get('clock_ghz')))"
  done
done 2>&1 | tee $O/ab_graded.txt
GPSIG_OPTIONS="sig_graded=1" python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400 | tee -a $O/ab_graded.txt
GPSIG_OPTIONS="sig_graded=0" python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400 | tee -a $O/ab_graded.txt
