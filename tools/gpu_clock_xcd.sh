#!/bin/bash
# per-XCD clock readings of the bench's probe on the default line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_clock.json
python - <<'P'
import json
d=json.loads(open('gpurun_out/bench_clock.json').read())
print('headline', round(d['ms_per_step'],3), d['roofline']['frac'], d['clock']['mean'], [(q['xcd'],round(q['ghz'],3)) for q in d['clock']['per_xcd']])
for s in d['secondary']:
    print(s['name'], round(s['ms_per_step'],3), s.get('kernel_ms'), s.get('issue_frac'), s.get('clock_ghz'), s.get('clock_ghz_xcd_min_max'))
P
python -m pytest tests/test_gpu_bench.py -q -m gpu -x 2>&1 | tail -3
