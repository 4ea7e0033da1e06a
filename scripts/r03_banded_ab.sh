#!/bin/bash
# Round 3, banded A/B on one MI355X: the funnel-shift loop (BGSA_BANDED_IMPL=a, round 2's default), the one-word-window
# loop with one group per wave (BGSA_BANDED_GROUPS=1) and with two (the default), every subject mix of config 3.
#   bash scripts/r03_banded_ab.sh <tag> [variants...]      variants: funnel cut_g1 cut_g2 (default: all)
out=gpurun_out/${1:-r03}; mkdir -p $out; shift
variants=${@:-"funnel cut_g1 cut_g2"}
run() { # run <tag> <env...>
  local tag=$1; shift
  env "$@" timeout -k 10 280 python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline --no-total > $out/banded_ab_$tag.json 2> $out/banded_ab_$tag.err
  echo "$tag rc=$?" | tee -a $out/banded_ab_summary.txt
  python - $out/banded_ab_$tag.json <<'PY' | tee -a $out/banded_ab_summary.txt
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  planted  %8.2f ms  %s" % (r["roofline"]["kernel_ms"], r["config"]["kernel"]))
    for n, v in r.get("banded_variants", {}).items():
        print("  %-9s %7.2f ms  kept %.2e" % (n, v["kernel_ms"], v["pairs_not_rejected_fraction"]))
except Exception as e:
    print("  no line:", e)
PY
}
for v in $variants; do
  case $v in
    funnel) run funnel BGSA_BANDED_IMPL=a;;
    cut_g1) run cut_g1 BGSA_BANDED_GROUPS=1;;
    cut_g2) run cut_g2 BGSA_BANDED_GROUPS=2;;
  esac
done
