mkdir -p gpurun_out/r02
for w in "" 2 4 8; do
for cfg in "--voxels 16384 --n-time 2048 --target-points 4096" "--voxels 32768 --n-time 1536 --target-points 1536" "--steps 60"; do
  XM_SEARCH_WORKERS=$w python bench.py $cfg --no-cpu-baseline --no-footnotes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('workers=$w', '$cfg', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],4), 'ms/step', d['speculation'])"
done; done > gpurun_out/r02/search_workers.txt 2>&1
