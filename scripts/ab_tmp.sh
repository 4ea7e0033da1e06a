set -e
run() { python3 bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print(r['config']['kernel'], '|', r['config']['length_bp'], '|', r['value'], 'GCUPS', r['roofline']['kernel_ms'],'ms')"; }
for wg in 512 256 768 1024 2048 4096; do
  echo "workgroups $wg"
  export BGSA_BLOCKED_WORKGROUPS=$wg
  run --config 2 --nq 200 --ns 20000 --length 4000
  BGSA_MYERS_MAX_PLAIN_WORDS=16 run --config 5
  run --config 4 --nq 200 --ns 64000 --length 1000
done
