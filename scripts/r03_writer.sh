#!/bin/bash
# The result-file writer of bgsa_amd/host/aligner on the full BASELINE size (10k x 1M x 150 bp, 20 GB of scores into
# /dev/shm): pwrite() with one stream (round 2) against the mapped file with 1 / 4 / 8 / 16 copy threads.  Every run's
# result file is compared with the first one's.
#   bash scripts/r03_writer.sh <tag> [queries] [subjects]
out=gpurun_out/${1:-r03}; mkdir -p $out
NQ=${2:-10000}; NS=${3:-1000000}; LEN=150
D=/dev/shm/bgsa_writer_$$; mkdir -p $D
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
run() { # run <label> <env...>
  local label=$1; shift
  echo "== $label" | tee -a $out/writer.txt
  ( cd $D && env "$@" timeout -k 10 300 $here/bgsa_amd/host/aligner -q query.txt -d subject.txt -f result_new.txt 2>&1 | grep -E "GCUPS|total time|cal_total|pipeline_busy|write_total|Error" ) | tee -a $out/writer.txt
  if [ -f $D/result_first.txt ]; then cmp $D/result_first.txt $D/result_new.txt && echo "result file identical to the first run's" | tee -a $out/writer.txt; rm -f $D/result_new.txt
  else mv $D/result_new.txt $D/result_first.txt; fi
}
MODES=${MODES:-pwrite mmap1 mmap4 mmap8 mmap16 populate8}
for m in $MODES; do
  case $m in
    pwrite) run "pwrite, 1 stream (default)" BGSA_WRITER_MODE=pwrite;;
    mmap*) run "mmap, ${m#mmap} copy thread(s)" BGSA_WRITER_MODE=mmap BGSA_WRITER_THREADS=${m#mmap};;
    falloc*) run "mmap + fallocate() ahead on a thread of its own, ${m#falloc} copy thread(s)" BGSA_WRITER_MODE=mmap+falloc BGSA_WRITER_THREADS=${m#falloc};;
    populate*) run "mmap + MADV_POPULATE_WRITE, ${m#populate} copy thread(s)" BGSA_WRITER_MODE=mmap+populate BGSA_WRITER_THREADS=${m#populate};;
  esac
done
rm -rf $D
