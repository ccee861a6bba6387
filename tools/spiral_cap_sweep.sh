#!/bin/bash
# sweep the level cap of the spiral's latency schedule (GG_SPIRAL_CAPS) at several batch sizes
for cap in 128 192 256 320 384 512 1024; do
  for b in 1 64 256; do
    echo -n "cap0=$cap batch=$b: "
    GG_SPIRAL_CAPS=$cap,64 timeout 120 python bench.py --cpu-seconds 0 --batch $b --steps 10 --warmup 3 | python tools/bench_brief.py
  done
done
