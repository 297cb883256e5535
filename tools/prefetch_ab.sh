#!/bin/bash
# the driver's exact headline command, 8 times each with PACOH_PREFETCH=0 and =1, alternating -> gpurun_out/prefetch_ab.txt
out=gpurun_out/prefetch_ab.txt
: > $out
for i in 1 2 3 4 5 6 7 8; do
  for p in 0 1; do
    line=$(PACOH_PREFETCH=$p python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline 2>/dev/null | tail -1)
    echo "$line" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('PACOH_PREFETCH=$p run $i: ms_per_step %.4f value %.1f host_ms_per_step %.4f' % (d['ms_per_step'], d['value'], d['host_ms_per_step']))" >> $out
  done
done
cat $out
