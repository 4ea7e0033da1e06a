#!/bin/bash
# Round 4: does the 150 bp Myers kernel (0.98 of the issue peak at the clock it sustains, eight waves per SIMD) hold a higher
# clock — and finish sooner — with fewer waves per SIMD?  Unused dynamic LDS caps the workgroups per CU.  Same box.
out=${1:-gpurun_out/r04/myers_occupancy.txt}
for pad in 0 20480 27000 32768 40960 54000; do
  r=$(BGSA_MYERS_LDS_PAD=$pad timeout -k 10 200 python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline --no-total --no-other-configs 2>/dev/null |
      python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['roofline']['kernel_ms'], r['value'], (r.get('clock') or {}).get('sustained_mhz'), r['checksum'])" 2>/dev/null)
  echo "LDS pad $pad B per workgroup: $r" | tee -a $out
done
