cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rnd in 1 2; do for pad in 1 2 3 4 8 9 16 17; do python tools/bench_lr.py --config c3 --pad $pad --verify 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('pad $pad', 'seq_features_ms %.3f'%d['stages']['seq_features_ms'], 'err %.1e'%d['rel_err_vs_oracle_same_randomness'])"; done; done
