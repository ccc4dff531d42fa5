#!/bin/bash
# usage (GPU box): bash scripts/box_spread.sh [runs=5] -- the driver's command several times on this allocation
nproc_q=$(cat /sys/fs/cgroup/cpu.max 2>/dev/null)
echo "box $(hostname) cpu.max '$nproc_q' $(date -u +%H:%M:%S)"
for i in $(seq 1 ${1:-5}); do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-footnotes --no-configs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['breakdown_ms']
print('  %.2f M spectra/s  %.4f ms/step  main kernel %.4f ms (%.3f of peak)  guess %.4f  selection %.4f  period median %.3f  searches started twice %d' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], b['guess_kernel'], b['selection_stage_kernels'], b['device_period_min_median_max'][1], d['speculation'].get('searches_started_twice', 0)), ' period max %.2f  search latency %.2f ms  throttled %s  cores %.1f' % (b['device_period_min_median_max'][2], b.get('search_latency_exchange_to_use', float('nan')), d['host_noise_timed_region'].get('cgroup_nr_throttled'), d['host_cores_used_rank0']))"
done
