#!/bin/bash
# GPU box, round 5: what the exact stale-range check (default for ranges <= 8 MiB) costs the two drop-ins on a bucket just under the limit.
cd "$(dirname "$0")/.."
D=/dev/shm/bgsa_strict_$$; mkdir -p $D
python3 - <<PY
import numpy as np
rng = np.random.default_rng(1)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
for name, n in (("query", 10000), ("subject", 64000)):
    rows = np.full((n, 151), 10, dtype=np.uint8)
    rows[:, :150] = acgt[rng.integers(0, 4, (n, 150))]
    rows.tofile("$D/" + name + ".txt")
PY
here=$(pwd)
for mode in "" -1 1; do
  for seam in original_hip original_hip_coarse; do
    echo "== $seam, BGSA_HIP_STRICT_RESIDENT='$mode' (10k x 64k x 150 bp: a 6.4 MB bucket) =="
    ( cd $D && t0=$(date +%s%N) && BGSA_HIP_STRICT_RESIDENT=$mode timeout -k 10 300 $here/oracle/_ref/$seam/aligner -q query.txt -d subject.txt -f r.txt -N 16 2>&1 | grep -E "cal_total|total time" | tr '\n' ' '; echo "wall $(( ($(date +%s%N) - t0) / 1000000 )) ms" )
  done
done
rm -rf $D
