# Same-box A/B of the search engines on the headline workload (bench.py --steps 20 --warmup 5): the host engine, the
# compute partition alone (no search kernel), the device engine with 8 / 4 reserved CUs and without a partition.
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline > gpurun_out/ab_$name.json 2>> gpurun_out/ab.err || echo FAIL $name; }
run host XMRIS_AMD_SEARCH=host
run host_part8 XMRIS_AMD_SEARCH=host XM_FORCE_PARTITION=8
run host_part4 XMRIS_AMD_SEARCH=host XM_FORCE_PARTITION=4
run dev8 XMRIS_AMD_SEARCH=device XM_SEARCH_CUS=8
run dev4 XMRIS_AMD_SEARCH=device XM_SEARCH_CUS=4
run dev_nopart XMRIS_AMD_SEARCH=device XM_SEARCH_PARTITION=0
run host2 XMRIS_AMD_SEARCH=host
python - <<PY
import json
for f in ("host","host_part8","host_part4","dev8","dev4","dev_nopart","host2"):
    try:
        d=json.loads(open("gpurun_out/ab_%s.json"%f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no result", e); continue
    b=d["breakdown_ms"]
    print("%-18s"%f, "value %.2f M"%(d["value"]/1e6), "ms/step %.4f"%d["ms_per_step"], "main %.4f"%b["main_kernel"], "guess %.4f"%b["guess_kernel"], "period med %.3f max %.3f"%(b["device_period_min_median_max"][1], b["device_period_min_median_max"][2]), "search lat %.2f gen %.2f"%(b["search_latency_exchange_to_use"], b["solver_generations"]), "cores %.1f"%d["host_cores_used_rank0"])
PY
