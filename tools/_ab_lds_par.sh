# A/B of one library build against the committed one on the SAME box: $1 = path of the other libcoalign_hip.so
for rep in 1 2; do
  for lib in new old; do
    if [ $lib = old ]; then cp coalign_amd/lib/libcoalign_hip.so /tmp/new.so; cp $1 coalign_amd/lib/libcoalign_hip.so; fi
    echo "== $lib rep $rep"; timeout 200 python tools/bench_conv_sp.py 2>/dev/null | tail -7 | cut -c1-230
    if [ $lib = old ]; then cp /tmp/new.so coalign_amd/lib/libcoalign_hip.so; fi
  done
done
