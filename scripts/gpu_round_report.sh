#!/bin/bash
# GPU record for profiles/: the GPU tests, the four BASELINE configs with CPU baselines, rocprofv3 kernel
# stats and PMC passes per config (separate --pmc runs, as MI355X_MICROARCH.md §HBM prescribes).
# Run on the GPU box from the repo root:  bash scripts/gpu_round_report.sh <tag> [parts]
#   parts: any of  tests bench prof pmc banded k31 ubench  (default: all);  CONFIGS="4" limits bench/prof/pmc to config 4
tag=${1:-r04}
parts=${2:-"tests bench prof pmc banded"}
CONFIGS=${CONFIGS:-"2 3 4 5"}      # restrict the bench / prof / pmc parts to some configs
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
has() { case " $parts " in *" $1 "*) return 0;; *) return 1;; esac; }
# a step that had to be killed says something about the GPU: stop, do not start the next one
guard() { if [ "$1" = 124 ] || [ "$1" = 137 ]; then echo "step killed (rc=$1): stopping" >> $out/summary.txt; exit 1; fi; }
step() { # step <name> <seconds> <cmd...>  -> runs under timeout, logs rc
  local name=$1 secs=$2; shift 2
  timeout -k 10 $secs "$@"; local rc=$?
  echo "$name rc=$rc" >> $out/summary.txt; echo "$name rc=$rc" >&2; guard $rc
}
if has tests; then
  step pytest 1000 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1
  tail -n 3 $out/pytest_gpu.log
fi
if has bench; then
  cfg() { case " $CONFIGS " in *" $1 "*) return 0;; *) return 1;; esac; }
  cfg 2 && step cfg2 500 python bench.py --config 2 > $out/bench_cfg2.json 2> $out/bench_cfg2.err
  cfg 3 && step cfg3 500 python bench.py --config 3 --cpu-sample 1000x100000 > $out/bench_cfg3.json 2> $out/bench_cfg3.err
  cfg 4 && step cfg4 600 python bench.py --config 4 --cpu-sample 1000x50000 > $out/bench_cfg4.json 2> $out/bench_cfg4.err
  cfg 5 && step cfg5 600 python bench.py --config 5 --steps 2 --cpu-sample 200x20000 > $out/bench_cfg5.json 2> $out/bench_cfg5.err
fi
if has prof; then
  for c in $CONFIGS; do
    step "prof cfg$c" 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_cfg$c -- python3 bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-total --no-clock-probe --no-other-configs --banded-variants '' > $out/prof_cfg$c.log 2>&1
  done
fi
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU"
if has pmc; then
  for c in $CONFIGS; do
    [ $c = 3 ] && continue    # config 3: the `banded` part, per subject mix
    for grp in FETCH_SIZE WRITE_SIZE SQ; do
      ctrs=$grp; [ $grp = SQ ] && ctrs=$SQ
      step "pmc cfg$c $grp" 400 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pmc_cfg${c}_$grp -- python3 bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline --no-total --no-clock-probe --no-other-configs > $out/pmc_cfg${c}_$grp.log 2>&1
    done
  done
fi
if has banded; then
  for mix in planted random dense1pct survivors; do
    for grp in FETCH_SIZE WRITE_SIZE SQ; do
      ctrs=$grp; [ $grp = SQ ] && ctrs=$SQ
      step "pmc cfg3 $mix $grp" 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pmc_cfg3_${mix}_$grp -- python3 bench.py --config 3 --banded-mix $mix --banded-variants '' --steps 1 --warmup 1 --no-cpu-baseline --no-total --no-clock-probe > $out/pmc_cfg3_${mix}_$grp.log 2>&1
    done
  done
fi
if has k31; then   # the reference's DEFAULT banded threshold (banded/BGSA_CPU/main.c:43), every pair surviving: time + the three PMC passes
  step "bench cfg3 k31" 300 python bench.py --config 3 --k 31 --banded-mix survivors --banded-variants '' --steps 3 --no-cpu-baseline --no-total > $out/bench_cfg3_k31.json 2> $out/bench_cfg3_k31.err
  for grp in FETCH_SIZE WRITE_SIZE SQ; do
    ctrs=$grp; [ $grp = SQ ] && ctrs=$SQ
    step "pmc cfg3 k31 $grp" 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pmc_cfg3_k31_$grp -- python3 bench.py --config 3 --k 31 --banded-mix survivors --banded-variants '' --steps 1 --warmup 1 --no-cpu-baseline --no-total --no-clock-probe > $out/pmc_cfg3_k31_$grp.log 2>&1
  done
fi
if has ubench; then
  ( cd scripts/ubench && ./valu_rate 8 2000 > ../../$out/ubench_valu_rate.txt 2>&1; ./body_rate 20000 > ../../$out/ubench_body_rate.txt 2>&1; ./bank_conflict 4 4000 > ../../$out/ubench_operand_cost.txt 2>&1; [ -x ./banded_mix ] && ./banded_mix 3000 > ../../$out/ubench_banded_mix.txt 2>&1 )
  step bitpal_sets 400 bash scripts/bitpal_sets_bench.sh $tag > $out/bitpal_sets.log 2>&1; cp gpurun_out/bitpal_sets_$tag.jsonl $out/bitpal_sets.jsonl 2>/dev/null
fi
echo done | tee -a $out/summary.txt
