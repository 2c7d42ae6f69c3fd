#!/bin/bash
# round 6: the wide route's tests, then the reference's shapes on every route
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wide.py -q -x -m gpu 2>&1 | tail -15 | cut -c1-400 | tee $O/pytest_wide.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_grad.py -q -x -m gpu -k "witness or weighted_tensor_vs_sequence or beyond_64 or wider_than_64" 2>&1 | tail -5 | cut -c1-400 | tee -a $O/pytest_wide.txt
for ds in ${DATASETS:-NetFlow Wafer JapaneseVowels ArabicDigits AUSLAN CMUsubject16 PEMS}; do
  timeout 400 python tools/reference_shapes.py $ds --routes ${ROUTES:-auto,matrix,wide} --reps 5 2>&1 | grep -v amdgpu | tee -a $O/reference_shapes.jsonl | cut -c1-420
done
