#!/bin/bash
# The reference's own host pipeline on the library (oracle/_ref/original_hip/aligner, align_hip seam): rows scored per launch
# (BGSA_HIP_ROW_AHEAD) and OpenMP threads, 10k x 1M x 150 bp into /dev/shm.
out=gpurun_out/${1:-r03}; mkdir -p $out
NQ=10000; NS=1000000; LEN=150
D=/dev/shm/bgsa_rowahead_$$; mkdir -p $D
python3 - <<PY
import numpy as np
rng = np.random.default_rng(1)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
for name, n in (("query", $NQ), ("subject", $NS)):
    rows = np.full((n, $LEN + 1), 10, dtype=np.uint8)
    rows[:, :$LEN] = acgt[rng.integers(0, 4, (n, $LEN))]
    rows.tofile("$D/" + name + ".txt")
PY
here=$(pwd)
for threads in ${THREADS:-16 32}; do for ahead in ${AHEADS:-16 32 64}; do
  echo "== -N $threads, BGSA_HIP_ROW_AHEAD=$ahead" | tee -a $out/rowahead.txt
  ( cd $D && BGSA_HIP_ROW_AHEAD=$ahead timeout -k 10 300 $here/oracle/_ref/original_hip/aligner -q query.txt -d subject.txt -f result.txt -N $threads 2>&1 | grep -E "GCUPS|total time|cal_total|write_total|Error|bgsa_hip|seam" ) | tee -a $out/rowahead.txt
  rm -f $D/result.txt*
done; done
rm -rf $D
