#!/bin/bash
cd "$(dirname "$0")"
[ -x lds_power ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 lds_power.hip -o lds_power
for mode in 1 0; do
  ./lds_power $mode 4 &
  pid=$!
  sleep 1.2
  for i in 1 2 3 4 5 6 7 8; do rocm-smi --showpower --showclocks --json 2>/dev/null | python3 -c "
import json,sys,re
d=json.load(sys.stdin); c=next(iter(d.values()))
pw=next((v for k,v in c.items() if 'power' in k.lower() and 'W' in k), None); sc=next((v for k,v in c.items() if 'sclk' in k.lower()), None)
print('   mode $mode power', pw, 'sclk', sc)"; sleep 0.25; done
  wait $pid
done
