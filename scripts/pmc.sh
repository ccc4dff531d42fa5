#!/bin/bash
# usage: scripts/pmc.sh <tag> <VARIANT> "<counters pass 1>" "<counters pass 2>" ...
# One rocprofv3 run per counter group (separate --pmc passes, as the MI355X guide prescribes).
tag=$1; var=$2; shift 2
export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  echo "pmc pass $i: $grp" >&2
  VARIANT=$var timeout -k 5 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$i -- python3 scripts/run_variant.py > /dev/null 2> gpurun_out/pmc_${tag}_$i.err || { tail -5 gpurun_out/pmc_${tag}_$i.err; exit 1; }
done
python3 scripts/pmc_summary.py gpurun_out/pmc_${tag}_*/ 
