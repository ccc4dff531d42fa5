# BASELINE configs[1] / configs[4] end to end (host engine) over searches in flight x team budget.
for w in 4 6 8; do for th in "" 8 16; do
  name="w${w}_t${th:-auto}"
  if [ -n "$th" ]; then export XM_SOLVER_THREADS=$th; else unset XM_SOLVER_THREADS; fi
  XM_SEARCH_WORKERS=$w timeout -k 10 200 python bench.py --only-configs --no-cpu-baseline > gpurun_out/sw_$name.json 2>> gpurun_out/sw.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/sw_$name.json").read().strip().splitlines()[-1])["configs"]
print("$name", {k.split()[0]: round(v["search_host"]["ms_per_dataset"],4) for k,v in d.items()})
PY
done; done
