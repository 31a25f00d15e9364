for a in 0 1 2 3 4 8 12 15; do echo "ablate $a: $(COALIGN_S2_ABLATE=$a timeout 100 python tools/bench_conv_s2.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:v['sp_s2_us'] for k,v in d.items()})")"; done
