#!/bin/bash
# GPU box, round 5: the reference's unmodified host pipeline on the library through BOTH seams, same synthetic files in /dev/shm, each
# printing the reference's own report — oracle/_ref/original_hip/aligner (fine seam: align_hip from its OpenMP grid, row cache) against
# oracle/_ref/original_hip_coarse/aligner (coarse seam: cpu_cal -> hip_cal_align_score, one device launch per 100-query block).
#   bash scripts/r05_coarse_drop_in.sh [queries] [subjects] [threads] > gpurun_out/r05_coarse_drop_in.txt
NQ=${1:-10000}; NS=${2:-1000000}; THREADS=${3:-16}; LEN=150
D=/dev/shm/bgsa_coarse_$$; mkdir -p $D
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
for seam in original_hip original_hip_coarse; do
  echo "== $seam/aligner -N $THREADS ($NQ x $NS x $LEN bp) =="
  ( cd $D && t0=$(date +%s%N) && BGSA_HIP_SEAM_STATS=1 timeout -k 10 600 $here/oracle/_ref/$seam/aligner -q query.txt -d subject.txt -f result_$seam.txt -N $THREADS 2>&1 | grep -E "GCUPS|total time|cal_total|Error|bgsa_hip" | cut -c1-260; echo "wall $(( ($(date +%s%N) - t0) / 1000000 )) ms" )
done
cmp $D/result_original_hip.txt $D/result_original_hip_coarse.txt && echo "result files identical"
rm -rf $D
