cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in 0 1 2 3 4 5; do python tools/bench_lr.py --config c3 --variant $v --verify 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('variant $v', 'seq_features_ms %.3f'%d['stages']['seq_features_ms'], 'total %.3f'%d['ms_per_evaluation'], 'err', d.get('rel_err_vs_oracle_same_randomness'))"; done
for v in 0 3 4; do python tools/bench_lr.py --config c2 --variant $v 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('c2 variant $v', 'seq_features_ms %.3f'%d['stages']['seq_features_ms'], 'total %.3f'%d['ms_per_evaluation'])"; done
