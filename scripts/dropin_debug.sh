D=/dev/shm/bgsa_dbg_$$; mkdir -p $D
python3 - <<PY
import numpy as np
rng = np.random.default_rng(1)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
for name, n in (("query", 1000), ("subject", 1000000)):
    rows = np.full((n, 151), 10, dtype=np.uint8)
    rows[:, :150] = acgt[rng.integers(0, 4, (n, 150))]
    rows.tofile("$D/" + name + ".txt")
PY
here=$(pwd)
cd $D && ( BGSA_HIP_SEAM_STATS=1 $here/oracle/_ref/original_hip/aligner -q query.txt -d subject.txt -f r.txt -N 1 > $D/out.txt 2> $D/err.txt; tail -4 $D/out.txt; tail -3 $D/err.txt )
rm -rf $D
