#!/bin/bash
# Round 5, final tree (the same sweeps as scripts/r04_final_sweeps.sh): the records that README's secondary rows cite — kernel rate by read length, the
# semi-global modes, the host-buffer seam host to host.  Output under gpurun_out/r05/.
out=gpurun_out/r05; mkdir -p $out
bash scripts/length_sweep.sh $out/length_sweep.txt > /dev/null 2>&1
echo "length sweep done"
timeout -k 10 300 python3 scripts/semi_perf.py > $out/semi_perf.txt 2>&1
echo "semi perf done"
timeout -k 10 200 python3 scripts/measure_host_path.py 2>/dev/null > $out/host_path.txt
echo "host path done"
for L in 4000 10000; do
  timeout -k 10 300 python3 bench.py --config 2 --length $L --nq 300 --ns 200000 --steps 2 --no-cpu-baseline --no-total --no-clock-probe 2>/dev/null |
    python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('myers $L bp (300 x 200k):', r['config']['kernel'], r['roofline']['kernel_gcups'], 'GCUPS')" >> $out/length_sweep.txt
done
tail -5 $out/length_sweep.txt; tail -6 $out/semi_perf.txt; tail -4 $out/host_path.txt
