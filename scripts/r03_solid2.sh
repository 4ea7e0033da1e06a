out=gpurun_out/r03j; mkdir -p $out
b3() { timeout -k 10 250 python bench.py --config 3 --banded-mix $1 --banded-variants '' --steps 3 --warmup 1 --no-cpu-baseline --no-total --no-clock-probe 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['roofline']['kernel_ms'])" 2>/dev/null; }
for cfg in "32 5 8" "32 5 16" "32 5 32" "24 5 16" "16 5 16" "24 5 32" "16 6 32"; do
  set -- $cfg
  line="push_solid=k+$1 margin=$2 push_max=$3:"
  for mix in random dense1pct planted; do line="$line $mix=$(BGSA_BANDED_PUSH_SOLID=$1 BGSA_BANDED_SOLID_MARGIN=$2 BGSA_BANDED_PUSH_MAX=$3 b3 $mix)"; done
  echo "$line" | tee -a $out/solid2.txt
done
