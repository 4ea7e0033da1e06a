#!/bin/bash
# Dynamic task handout (bgsa_common.h) against the static grids, same box: kernel ms per BASELINE config.
out=gpurun_out/${1:-r03}; mkdir -p $out
one() { # one <label> <env> <bench args...>
  local label=$1 e=$2; shift 2
  r=$(env $e timeout -k 10 400 python bench.py "$@" --no-cpu-baseline --no-total --banded-variants '' 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['roofline']['kernel_ms'], r['value'], r['config']['kernel'], r['checksum'])" 2>/dev/null)
  echo "$label: ${r:-fail}" | tee -a $out/dynamic_ab.txt
}
for rep in 1 2; do
one "cfg2 static " BGSA_DYNAMIC_TASKS=0 --config 2 --steps 4
one "cfg2 dynamic" BGSA_DYNAMIC_TASKS=1 --config 2 --steps 4
one "cfg5 static " BGSA_DYNAMIC_TASKS=0 --config 5 --steps 2
one "cfg5 dynamic" BGSA_DYNAMIC_TASKS=1 --config 5 --steps 2
one "cfg4 static " BGSA_DYNAMIC_TASKS=0 --config 4 --steps 1 --nq 2000
one "cfg4 dynamic" BGSA_DYNAMIC_TASKS=1 --config 4 --steps 1 --nq 2000
one "cfg3 static " BGSA_BANDED_DYNAMIC=0 --config 3 --steps 5 --banded-mix survivors
one "cfg3 dynamic" BGSA_BANDED_DYNAMIC=1 --config 3 --steps 5 --banded-mix survivors
one "cfg3 random static " BGSA_BANDED_DYNAMIC=0 --config 3 --steps 5 --banded-mix random
one "cfg3 random dynamic" BGSA_BANDED_DYNAMIC=1 --config 3 --steps 5 --banded-mix random
done
for len in 64 250; do
one "myers $len static " BGSA_DYNAMIC_TASKS=0 --config 2 --length $len --nq 3000 --steps 2
one "myers $len dynamic" BGSA_DYNAMIC_TASKS=1 --config 2 --length $len --nq 3000 --steps 2
done
