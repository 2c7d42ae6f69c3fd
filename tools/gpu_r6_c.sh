#!/bin/bash
# round 6: the wide route's tests (Kzx + sequence lattices), the tests it could have disturbed, the reference's shapes per primitive
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_wide.py -q -x -m gpu 2>&1 | tail -15 | cut -c1-600 | tee $O/pytest_wide.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_grad.py -q -x -m gpu -k "witness or weighted_tensor_vs_sequence or beyond_64 or wider_than_64 or wave_and_storage or seq_level_gradients" 2>&1 | tail -5 | cut -c1-600 | tee -a $O/pytest_wide.txt
for ds in ${DATASETS:-NetFlow Wafer JapaneseVowels ArabicDigits AUSLAN CMUsubject16 PEMS}; do
  timeout 400 python tools/reference_shapes.py $ds --routes ${ROUTES:-auto,matrix,wide} --reps 5 2>&1 | grep -v amdgpu | tee -a $O/reference_shapes.jsonl | cut -c1-900
done
