# A/B on one box: low-rank sequence features, two-array fused kernel (--fused 1) against the three-array one (--fused 2), alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "low_rank" 2>&1 | tail -2
for rnd in 1 2 3; do for cfg in "c3" "c2" "c3 --sparsity log"; do for f in 1 2; do python tools/bench_lr.py --config $cfg --fused $f --verify 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('round $rnd  $cfg  fused=$f', 'seq_features_ms %.3f'%d['stages']['seq_features_ms'], 'eval %.3f'%d['ms_per_evaluation'], 'err %.1e'%d['rel_err_vs_oracle_same_randomness'])"; done; done; done
