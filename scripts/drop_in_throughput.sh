#!/bin/bash
# What a BGSA maintainer sees after dropping the library in (GPU box): the reference's OWN host pipeline
# (main.c / file.c / thread.c / cal_cpu.c, unmodified, oracle/_ref/original_hip/aligner) and this repo's
# command line, on the same synthetic files, each printing the reference-style report (cal / Total GCUPS).
#   bash scripts/drop_in_throughput.sh [queries] [subjects] [length] [host threads] [myers|banded]
# banded: banded/BGSA_CPU's host files on the library (oracle/_ref/banded_hip/aligner -k 8, int8 scores) against `aligner -a banded -k 8`
NQ=${1:-1000}; NS=${2:-1000000}; LEN=${3:-150}; THREADS=${4:-16}; MODE=${5:-myers}
REFBIN=oracle/_ref/original_hip/aligner; REFARGS=""; OURARGS=""
if [ "$MODE" = banded ]; then REFBIN=oracle/_ref/banded_hip/aligner; REFARGS="-k 8"; OURARGS="-a banded -k 8"; fi
D=/dev/shm/bgsa_dropin_$$; mkdir -p $D
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
echo "== reference host files on libbgsa_hip.so (align_hip seam from its OpenMP loop), -N $THREADS =="
( cd $D && t0=$(date +%s%N) && timeout -k 10 900 $here/$REFBIN -q query.txt -d subject.txt -f result_ref.txt -N $THREADS $REFARGS 2>&1 | grep -E "GCUPS|total time|cal_total|Error|bgsa_hip"; echo "wall $(( ($(date +%s%N) - t0) / 1000000 )) ms" )
echo "== bgsa_amd/host/aligner (device-resident pipeline on the same C ABI) =="
( cd $D && t0=$(date +%s%N) && timeout -k 10 900 $here/bgsa_amd/host/aligner -q query.txt -d subject.txt -f result_hip.txt $OURARGS 2>&1 | grep -E "GCUPS|total time|cal_total|pipeline_busy|write_total|Error|bgsa_hip"; echo "wall $(( ($(date +%s%N) - t0) / 1000000 )) ms" )
cmp $D/result_ref.txt $D/result_hip.txt && echo "result files identical"
echo "== the same with the result sent to /dev/null: what the pipeline does when the sink keeps up (no file is kept: not a Total GCUPS) =="
( cd $D && ln -s /dev/null sink.txt && timeout -k 10 900 $here/bgsa_amd/host/aligner -q query.txt -d subject.txt -f sink.txt $OURARGS 2>&1 | grep -E "GCUPS|total time|cal_total|pipeline_busy|write_total|Error" )
rm -rf $D
