set -e
run() { python3 bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print(r['config']['kernel'], '|', r['config']['queries'],'x',r['config']['subjects_per_gpu'],'x',r['config']['length_bp'], '|', r['value'], 'GCUPS')"; }
for i in 1 2; do
echo sched; run --config 4 --nq 2000
echo nosched; BGSA_HIP_LIB=$PWD/bgsa_amd/libbgsa_hip_nosched.so run --config 4 --nq 2000
done
echo sched250; run --config 4 --nq 1000 --ns 256000 --length 250
echo nosched250; BGSA_HIP_LIB=$PWD/bgsa_amd/libbgsa_hip_nosched.so run --config 4 --nq 1000 --ns 256000 --length 250
echo sched64; run --config 4 --nq 2000 --ns 1000000 --length 64
echo nosched64; BGSA_HIP_LIB=$PWD/bgsa_amd/libbgsa_hip_nosched.so run --config 4 --nq 2000 --ns 1000000 --length 64
