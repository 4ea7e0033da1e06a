#!/bin/bash
# GPU box: BitPAl throughput of every compiled score set (2k queries x 1M subjects x 150 bp), plus the
# BASELINE config 4 line for the default set.  Output: gpurun_out/bitpal_sets_<tag>.jsonl
set -e
TAG=${1:-r01}
OUT=gpurun_out/bitpal_sets_$TAG.jsonl
: > $OUT
python3 bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-total >> $OUT
for S in $(python3 -c "import bgsa_amd as B; print(' '.join(','.join(map(str,s)) for s in B.score_sets()))"); do
    python3 bench.py --config 4 --scores=$S --nq 2000 --steps 2 --warmup 1 --no-cpu-baseline --no-total >> $OUT
    python3 bench.py --config 4 --scores=$S --nq 200 --ns 64000 --length 1000 --steps 2 --warmup 1 --no-cpu-baseline --no-total >> $OUT
done
python3 - <<PY
import json
for line in open("$OUT"):
    r = json.loads(line)
    print(r["config"]["workload"][:60], "|", r["config"]["kernel"], "|", r["value"], "GCUPS |", r["roofline"]["issued"])
PY
