#!/bin/bash
out=gpurun_out/r03y; mkdir -p $out
NQ=10000; NS=1000000; LEN=150
D=/dev/shm/bgsa_st_$$; mkdir -p $D
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
for ahead in ${AHEADS:-100 107 64}; do
  echo "== BGSA_HIP_ROW_AHEAD=$ahead" | tee -a $out/stats.txt
  ( cd $D && BGSA_HIP_SEAM_STATS=1 BGSA_HIP_ROW_AHEAD=$ahead timeout -k 10 300 $here/oracle/_ref/original_hip/${BIN:-aligner} -q query.txt -d subject.txt -f result.txt -N 16 2>&1 | grep -v "^$" | grep -E "bgsa_hip|cal_total|cal GCUPS|Total GCUPS" ) | tee -a $out/stats.txt
  rm -f $D/result.txt*
done
rm -rf $D
