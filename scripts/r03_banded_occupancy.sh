#!/bin/bash
out=gpurun_out/r03o; mkdir -p $out
for mix in random survivors dense1pct; do
for pad in 0 34000 50000 70000; do
  r=$(BGSA_BANDED_LDS_PAD=$pad timeout -k 10 200 python bench.py --config 3 --steps 5 --banded-mix $mix --no-cpu-baseline --no-total --no-clock-probe --banded-variants '' 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['roofline']['kernel_ms'], r['checksum'])")
  echo "$mix pad=$pad: $r" | tee -a $out/banded_occupancy.txt
done; done
