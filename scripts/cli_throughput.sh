#!/bin/bash
# End-to-end run of the C command line on synthetic files (GPU box): prints the reference-style report.
set -e
NQ=${1:-1000}; NS=${2:-1000000}; LEN=${3:-150}; ALGO=${4:-myers}
D=/dev/shm/bgsa_cli_$$; mkdir -p $D
python3 - <<PY
import numpy as np
rng = np.random.default_rng(1)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
for name, n in (("query", $NQ), ("subject", $NS)):
    rows = np.full((n, $LEN + 1), 10, dtype=np.uint8)
    rows[:, :$LEN] = acgt[rng.integers(0, 4, (n, $LEN))]
    rows.tofile("$D/" + name + ".txt")
PY
( cd $D && time $OLDPWD/bgsa_amd/host/aligner -q query.txt -d subject.txt -f result.txt -a $ALGO -k 8 )
ls -la $D | tail -4
rm -rf $D
