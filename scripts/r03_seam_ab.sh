#!/bin/bash
# Small launches (one tile of the coarse seam is 12 queries x 1M subjects): queries per task and the task handout.
# Kernel-only time by launch size: static grids, the task counter above its default floor (bgsa_common.h
# dynamic_min_tasks), the counter at every size, and the finer task target; then the seam itself.  One box.
out=gpurun_out/${1:-r03}; mkdir -p $out
one() { # one <label> <env> <bench args...>
  local label=$1 e=$2; shift 2
  r=$(env $e timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-total --no-clock-probe --banded-variants '' 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['roofline']['kernel_ms'], r['value'], r['config']['kernel'], r['checksum'])" 2>/dev/null)
  echo "$label: ${r:-fail}" | tee -a $out/seam_ab.txt
}
for nq in 12 25 100 200 400 1000; do
  one "150 bp nq=$nq static           " "BGSA_DYNAMIC_TASKS=0" --config 2 --length 150 --nq $nq --steps 20 --warmup 3
  one "150 bp nq=$nq default          " "BGSA_DYNAMIC_TASKS=1" --config 2 --length 150 --nq $nq --steps 20 --warmup 3
  one "150 bp nq=$nq counter always   " "BGSA_DYNAMIC_MIN_TASKS=1" --config 2 --length 150 --nq $nq --steps 20 --warmup 3
  one "150 bp nq=$nq counter, 1-word floor" "BGSA_DYNAMIC_MIN_TASKS=1 BGSA_DYNAMIC_TASK_WORDS=1" --config 2 --length 150 --nq $nq --steps 20 --warmup 3
done
for d in 0 1 0 1; do
  echo "seam tiles=8 dynamic=$d: $(BGSA_DYNAMIC_TASKS=$d timeout -k 10 200 python scripts/measure_host_path.py 2>&1 | tail -1 | sed 's/.*steady //; s/ host-to-host.*//')" | tee -a $out/seam_ab.txt
done
