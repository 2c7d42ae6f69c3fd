#!/bin/bash
# round 6: the reference's 16 shapes, auto route -- times per primitive + which kernels ran (one kernel trace per data set)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r06h; mkdir -p $O; rm -f $O/shapes.jsonl
for ds in DigitShapes Shapes ECG LIBRAS PenDigits CharacterTrajectories UWave NetFlow Wafer JapaneseVowels ArabicDigits AUSLAN CMUsubject16 KickvsPunch WalkvsRun PEMS; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$ds -o p -- python tools/reference_shapes.py $ds --routes auto --reps 3 > $O/run_$ds.log 2>&1
  db=$(find $O/prof_$ds -name '*.db' | head -1)
  python tools/rocprof_summary.py stats "$db" > $O/kernels_$ds.txt 2>&1
  rm -rf $O/prof_$ds
  grep '"dataset"' $O/run_$ds.log >> $O/shapes.jsonl
done
# un-profiled times (the trace costs a few per cent)
rm -f $O/shapes_clean.jsonl
for ds in DigitShapes Shapes ECG LIBRAS PenDigits CharacterTrajectories UWave NetFlow Wafer JapaneseVowels ArabicDigits AUSLAN CMUsubject16 KickvsPunch WalkvsRun PEMS; do
  timeout 300 python tools/reference_shapes.py $ds --routes auto --reps 5 2>/dev/null | grep '"dataset"' >> $O/shapes_clean.jsonl
done
wc -l $O/shapes.jsonl $O/shapes_clean.jsonl
