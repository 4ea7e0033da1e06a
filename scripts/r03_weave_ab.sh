#!/bin/bash
# A/B of the dispatch woven under the row body (gen_rows_asm.py: weave_dispatch): bgsa_amd/libbgsa_hip_noweave.so is the
# same tree built with the dispatch behind the body (round 2's form).  Kernel ms per config.
out=gpurun_out/${1:-r03}; mkdir -p $out
one() { # one <label> <lib or ""> <bench args...>
  local label=$1 lib=$2; shift 2
  ms=$(BGSA_HIP_LIB=$lib timeout -k 10 400 python bench.py "$@" --no-cpu-baseline --no-total --no-clock-probe --banded-variants '' 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['roofline']['kernel_ms'], r['config']['kernel'])" 2>/dev/null)
  echo "$label: ${ms:-fail}" | tee -a $out/weave_ab.txt
}
for rep in 1 2; do
one "cfg2 noweave" $PWD/bgsa_amd/libbgsa_hip_noweave.so --config 2 --steps 3
one "cfg2 weave  " "" --config 2 --steps 3
one "cfg5 noweave" $PWD/bgsa_amd/libbgsa_hip_noweave.so --config 5 --steps 2
one "cfg5 weave  " "" --config 5 --steps 2
one "cfg4 noweave" $PWD/bgsa_amd/libbgsa_hip_noweave.so --config 4 --steps 1 --nq 2000
one "cfg4 weave  " "" --config 4 --steps 1 --nq 2000
done
for len in 64 300 576 2000; do
one "myers $len noweave" $PWD/bgsa_amd/libbgsa_hip_noweave.so --config 2 --length $len --nq 2000 --steps 2
one "myers $len weave  " "" --config 2 --length $len --nq 2000 --steps 2
done
