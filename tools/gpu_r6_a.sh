#!/bin/bash
# round 6, first visit: the new tests (witness values, float32 capture refusal) and the reference's own shapes on every existing route
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06a; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_graph.py -q -x -m gpu -k "witness or float32_requests or replay_equals" 2>&1 | tail -5 | tee $O/pytest_new.txt
for ds in DigitShapes ECG LIBRAS PenDigits CharacterTrajectories UWave NetFlow Wafer JapaneseVowels ArabicDigits AUSLAN CMUsubject16 WalkvsRun PEMS; do
  timeout 400 python tools/reference_shapes.py $ds --routes auto,library,matrix --reps 5 2>&1 | grep -v amdgpu | tee -a $O/reference_shapes_before.jsonl | cut -c1-400
done
