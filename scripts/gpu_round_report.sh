#!/bin/bash
# Full GPU record for profiles/: tests, the four BASELINE configs with CPU baselines, rocprofv3 stats.
# Run on the GPU box from the repo root:  bash scripts/gpu_round_report.sh <tag>
tag=${1:-r01_v7}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
# a step that had to be killed says something about the GPU: stop, do not start the next one
guard() { if [ "$1" = 124 ] || [ "$1" = 137 ]; then echo "step killed (rc=$1): stopping" | tee -a $out/summary.txt; exit 1; fi; }
timeout -k 10 500 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $out/summary.txt; guard $rc
timeout -k 10 400 python bench.py --config 2 > $out/bench_cfg2.json 2> $out/bench_cfg2.err; rc=$?; echo "cfg2 rc=$rc" | tee -a $out/summary.txt; guard $rc
timeout -k 10 400 python bench.py --config 3 --cpu-sample 2000x100000 > $out/bench_cfg3.json 2> $out/bench_cfg3.err; rc=$?; echo "cfg3 rc=$rc" | tee -a $out/summary.txt; guard $rc
timeout -k 10 500 python bench.py --config 4 --cpu-sample 1000x50000 > $out/bench_cfg4.json 2> $out/bench_cfg4.err; rc=$?; echo "cfg4 rc=$rc" | tee -a $out/summary.txt; guard $rc
timeout -k 10 400 python bench.py --config 5 --cpu-sample 200x20000 > $out/bench_cfg5.json 2> $out/bench_cfg5.err; rc=$?; echo "cfg5 rc=$rc" | tee -a $out/summary.txt; guard $rc
for c in 2 3 4 5; do
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_cfg$c -- python3 bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > $out/prof_cfg$c.log 2>&1
  rc=$?; echo "prof cfg$c rc=$rc" | tee -a $out/summary.txt; guard $rc
done
# PMC passes for the headline config (separate runs, as MI355X_MICROARCH.md §HBM prescribes)
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc_$ctr -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/pmc_$ctr.log 2>&1
  rc=$?; echo "pmc $ctr rc=$rc" | tee -a $out/summary.txt; guard $rc
done
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d $out/pmc_SQ -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/pmc_SQ.log 2>&1
rc=$?; echo "pmc SQ rc=$rc" | tee -a $out/summary.txt; guard $rc
# issue-rate microbenchmarks behind DESIGN.md §4.1
( cd scripts/ubench && ./valu_rate 8 2000 > ../../$out/ubench_valu_rate.txt 2>&1; ./body_rate 20000 > ../../$out/ubench_body_rate.txt 2>&1; ./bank_conflict 4 4000 > ../../$out/ubench_operand_cost.txt 2>&1 )
# every compiled BitPAl score set
timeout -k 10 400 bash scripts/bitpal_sets_bench.sh $tag > $out/bitpal_sets.log 2>&1; cp gpurun_out/bitpal_sets_$tag.jsonl $out/bitpal_sets.jsonl 2>/dev/null
tail -c 300 $out/pytest_gpu.log
