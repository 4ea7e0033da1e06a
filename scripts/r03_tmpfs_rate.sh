#!/bin/bash
# What the box's /dev/shm takes, with nothing of ours in the way: dd from /dev/zero into ONE file, then into four and
# eight files at once (the single `result` file of the reference's format is what the writer of aligner.c is bound by).
out=gpurun_out/${1:-r03}; mkdir -p $out; D=/dev/shm/bgsa_rate_$$; mkdir -p $D
rate() { # rate <files> <GiB each>
  local n=$1 g=$2 t0=$(date +%s%N)
  for i in $(seq 1 $n); do dd if=/dev/zero of=$D/f$i bs=16M count=$((g * 64)) status=none & done; wait
  local ms=$(( ($(date +%s%N) - t0) / 1000000 ))
  echo "$n file(s) x $g GiB: $ms ms = $(python3 -c "print(round($n * $g * 1.073741824 / ($ms / 1000.0), 2))") GB/s" | tee -a $out/tmpfs_rate.txt
  rm -f $D/f*
}
nproc | sed 's/^/cpus: /' | tee -a $out/tmpfs_rate.txt
rate 1 16; rate 1 16; rate 2 8; rate 4 4; rate 8 2; rate 16 1
rmdir $D
