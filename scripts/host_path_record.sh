mkdir -p gpurun_out/r2x
python -m pytest tests/test_gpu_parity.py -x -q -k "copy_out_in_tiles or host_surface or resident" > gpurun_out/r2x/t.log 2>&1; tail -3 gpurun_out/r2x/t.log
python scripts/measure_host_path.py 2>/dev/null > gpurun_out/r2x/host_path.txt
for t in 1 2 4 6; do echo "BGSA_HIP_SEAM_TILES=$t:" >> gpurun_out/r2x/host_path.txt; BGSA_HIP_SEAM_TILES=$t python scripts/measure_host_path.py 2>/dev/null | tail -1 >> gpurun_out/r2x/host_path.txt; done
cat gpurun_out/r2x/host_path.txt
