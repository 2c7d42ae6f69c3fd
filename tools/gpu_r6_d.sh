#!/bin/bash
# round 6: the whole GPU suite with the wide route in place + the reference-shape sweep (all 16 data sets, auto route; the A/B routes for the wide ones)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06d; mkdir -p $O
timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -15 | cut -c1-600 | tee $O/pytest_gpu.txt
for ds in DigitShapes ECG LIBRAS PenDigits Shapes CharacterTrajectories UWave; do
  timeout 400 python tools/reference_shapes.py $ds --routes auto --reps 5 2>&1 | grep -v amdgpu | tee -a $O/reference_shapes.jsonl | cut -c1-300
done
for ds in NetFlow Wafer JapaneseVowels ArabicDigits AUSLAN CMUsubject16 KickvsPunch WalkvsRun PEMS; do
  timeout 400 python tools/reference_shapes.py $ds --routes auto,matrix --reps 5 2>&1 | grep -v amdgpu | tee -a $O/reference_shapes.jsonl | cut -c1-300
done
