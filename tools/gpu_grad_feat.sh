#!/bin/bash
# Round 4: SignatureLinear's Gram, forward + backward, through the feature route against the pair kernels' reverse pass.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04; mkdir -p $O
{
for shape in "1024 64 8 5" "4096 64 8 5" "1024 100 6 4" "4096 100 6 4" "1024 50 3 4" "512 200 4 4" "2048 64 16 3"; do
  set -- $shape
  for opt in "sig_features_grad=0" "sig_features_grad=-1"; do
    [ "$opt" = "sig_features_grad=0" ] && [ "$1" = "4096" ] && [ "$3" = "8" ] && continue      # (minutes on the pair kernels)
    GPSIG_OPTIONS="$opt" timeout 600 python tools/bench_grad_gram.py $1 linear 3 $2 $3 $4 2>&1 | tail -1
  done
done
} | tee $O/bench_grad_feat.txt
