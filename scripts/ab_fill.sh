#!/bin/bash
# usage (GPU box): bash scripts/ab_fill.sh [rounds=4] -- the pipeline fill of a run_stream call, driver command, ms per step:
#   A  one rank, event-driven fill (XM_FAST_FILL=1, the default)
#   B  one rank, ramped look-ahead (XM_FAST_FILL=0 XM_FILL_RAMP=1)
#   C  the MULTI-rank code path on one rank (XM_BENCH_SOLO_EXCHANGE=1), fixed order, whole look-ahead first (XM_FILL_RAMP=0)
#   D  the same with the ramped look-ahead (XM_FILL_RAMP=1)
run() {
  label=$1; shift
  v=$(env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-footnotes --no-configs 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 4))")
  echo -n "$label $v   "
}
for r in $(seq 1 ${1:-4}); do
  run A XM_FAST_FILL=1
  run B XM_FAST_FILL=0 XM_FILL_RAMP=1
  run C XM_BENCH_SOLO_EXCHANGE=1 XM_FILL_RAMP=0
  run D XM_BENCH_SOLO_EXCHANGE=1 XM_FILL_RAMP=1
  echo
done
