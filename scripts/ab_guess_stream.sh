#!/bin/bash
# usage (GPU box): bash scripts/ab_guess_stream.sh -- the guess + selection chain on the compute stream (default) against a
# stream of its own (XM_GUESS_STREAM=1): BASELINE configs[1] / [4] (ms per dataset, host engine) and the headline at K = 20.
for v in 0 1 0 1; do
  c=$(XM_GUESS_STREAM=$v python3 bench.py --only-configs --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())['configs']
print(' '.join('%s %.4f (main %.4f)' % (k.split()[0], v['search_host']['ms_per_dataset'], v['search_host']['main_kernel_ms']) for k,v in d.items()))")
  h=$(XM_GUESS_STREAM=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-footnotes --no-configs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline K=20 %.4f ms/step, main kernel %.4f ms' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))")
  echo "XM_GUESS_STREAM=$v  $c  |  $h"
done
