#!/bin/bash
# usage (GPU box): bash scripts/rehearse_ranks.sh [ranks=6] -- N ranks sharing the one GPU (process guard: at most six),
# the driver-shaped command, under the executor's switches one at a time: which of them the multi-rank step depends on.
n=${1:-6}
run() {
  label=$1; shift
  out=$(env "$@" python3 bench.py --gpus $n --share-gpu --dist-backend gloo --voxels 8192 --steps 20 --warmup 5 --no-cpu-baseline --no-footnotes 2>/dev/null | grep '^{')
  python3 - "$label" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
pr = d.get("per_rank") or []
lat = [round(r.get("search_latency_mean_ms", float("nan")), 2) for r in pr]
per = [round((r.get("device_period_median_ms") or float("nan")), 3) for r in pr]
print(f"{sys.argv[1]:28s} ms/step {d['ms_per_step']:9.3f}  value {d['value'] / 1e6:7.2f} M  device-engine searches {d.get('speculation', {}).get('searches_device_engine')}  period {per}  search latency {lat}")
PY
}
if [ -n "$XM_REHEARSE_SET" ]; then
  for spec in $XM_REHEARSE_SET; do run "$spec" ${spec//,/ }; done  # (commas join several assignments)
  exit 0
fi
run default XM_NOP=1
run polish_threads XM_POLISH_THREADS=1
run polish_native XMRIS_AMD_POLISH=native
run python_threads XM_SEARCH_PYTHON_THREADS=1
run no_hedge XMRIS_AMD_HEDGE=0
run device_engine XMRIS_AMD_SEARCH=device
