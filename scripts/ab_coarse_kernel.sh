#!/bin/bash
# usage (GPU box): bash scripts/ab_coarse_kernel.sh -- per-kernel means of a bench run (rocprofv3 --kernel-trace --stats) with
# the matrix-core coarse-spectra kernel (default) and with the FFT one (XM_GUESS_FFT=1)
export TMPDIR=/tmp
for v in mfma fft; do
  d=gpurun_out/ab_coarse_$v
  rm -rf $d
  if [ $v = fft ]; then export XM_GUESS_FFT=1; else unset XM_GUESS_FFT; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-footnotes --no-configs > /dev/null 2>&1
  echo "== $v"
  python3 - "$(find $d -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for row in list(csv.DictReader(open(sys.argv[1])))[:5]:
    print(f"  {row['Name'][:70]:70s} calls {row['Calls']:>5s}  mean {float(row['AverageNs']) / 1e3:8.1f} us")
PY
  rm -rf $d
done
