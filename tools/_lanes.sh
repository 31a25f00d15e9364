for rep in 1 2; do
for cfg in "2 3" "2 2" "3 2" "2 4" "3 1" "3 3"; do set -- $cfg
  for st in 40 20; do
  timeout 200 python bench.py --lanes $1 --queue-depth $2 --steps $st --warmup 5 --no-cpu-baseline --no-numerics --no-side-modes --no-from-points --no-latency --no-size-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('lanes $1 depth $2 steps $st rep $rep fps', d['value'])"
  done
done; done
