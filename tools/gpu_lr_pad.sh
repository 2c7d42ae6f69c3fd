# A/B on one box: row stride of the fused low-rank feature kernel's LDS arrays (64 * chunks + pad doubles), alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rnd in 1 2 3; do for pad in 1 0 2; do for f in 1 2; do python tools/bench_lr.py --config c3 --pad $pad --fused $f --verify 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('round $rnd pad $pad fused $f', 'seq_features_ms %.3f'%d['stages']['seq_features_ms'], 'err %.1e'%d['rel_err_vs_oracle_same_randomness'])"; done; done; done
