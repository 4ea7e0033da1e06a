#!/bin/bash
# How profiles/r05_* were produced (GPU box; each line was its own `gpurun` call, in this order, on the final tree):
#   CONFIGS="2 4 5" bash scripts/gpu_round_report.sh r05 "prof pmc"        # rocprofv3 kernel stats + three PMC passes per config
#   bash scripts/gpu_round_report.sh r05 "banded k31"                      # config 3: four subject mixes + the reference's default threshold
#   python3 scripts/collect_profiles.py gpurun_out/r05 r05 && git commit   # stamped passes in the tree BEFORE the bench lines are taken
#   python -m pytest tests -m gpu -q -x                                     # -> r05_pytest_gpu.log
#   python bench.py --steps 20 --warmup 5                                   # the driver's command -> r05_bench_driver_command.json (= r05_cfg2_bench.json)
#   CONFIGS="3 4 5" bash scripts/gpu_round_report.sh r05 "bench"            # -> r05_cfg{3,4,5}_bench.json
#   scripts/r05_scale_rehearsal.sh 4                                        # -> r05_scale_rehearsal.json (four ranks on one card, gloo: shape, not speed)
#   bash scripts/r05_final_sweeps.sh                                        # -> r05_length_sweep.txt, r05_semi_perf_lengths.txt, r05_host_path.txt
#   python3 scripts/soak_parity.py / soak_seams.py (arguments in the file)  # -> r05_soak.txt
# The A/B records (r05_split_*, r05_ilp_ab, r05_park_ab, r05_balance_ab, r05_tile_ab, r05_semi_perf, r05_dephase_ab, r05_block_rows) come from
# the scripts of the same names, with the measurement libraries built by scripts/build_variant.sh as each script's header says.
echo "this file is a record, not a driver: run the lines above one gpurun call at a time" >&2
exit 0
