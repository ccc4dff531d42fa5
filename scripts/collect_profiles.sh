#!/bin/bash
# usage (on the GPU box, from the repo root): bash scripts/collect_profiles.sh <round-tag>
# Regenerates everything under profiles/<round-tag>/ source data (written to gpurun_out/<round-tag>/):
#   1. bench.py unprofiled            -> bench_unprofiled.json
#   2. bench.py under rocprofv3 stats -> bench_under_rocprof.json, bench_kernel_stats.csv, bench_kernel_trace_xm_kernels.csv
#   3. separate --pmc passes for the main kernel (speculative schedule: write + phase + per-row maxima), the guess
#      kernel (xm_row_l1) and the classic schedule's pre-pass -> pmc_main_kernel.txt, pmc_guess_kernel.txt,
#      pmc_prepass_kernel.txt
set -e -o pipefail
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py --no-cpu-baseline 2> $out/bench_unprofiled.err | grep '^{' > $out/bench_unprofiled.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline \
    2> $out/bench_under_rocprof.err | grep '^{' > $out/bench_under_rocprof.json
f=$(find $out/stats -name '*kernel_stats.csv' | head -1)
cp "$f" $out/bench_kernel_stats.csv
t=$(find $out/stats -name '*kernel_trace.csv' | head -1)
{ head -1 "$t"; grep -E '"(void )?k_' "$t" || true; } > $out/bench_kernel_trace_xm_kernels.csv
rm -rf $out/stats
# small groups: a pass that asks for more counters than the hardware can collect at once aborts
PMC_GROUPS=("FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"
        "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
        "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum")
rm -rf gpurun_out/pmc_${tag}main_* gpurun_out/pmc_${tag}pre_*
bash scripts/pmc.sh ${tag}main all "${PMC_GROUPS[@]}" > $out/pmc_main_kernel.txt
echo "main-kernel counters done"
rm -rf gpurun_out/pmc_${tag}guess_*
bash scripts/pmc.sh ${tag}guess guess "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU" > $out/pmc_guess_kernel.txt
rm -rf gpurun_out/pmc_${tag}guess_*
bash scripts/pmc.sh ${tag}pre pre "${PMC_GROUPS[@]}" > $out/pmc_prepass_kernel.txt
rm -rf gpurun_out/pmc_${tag}main_* gpurun_out/pmc_${tag}pre_*
echo "collected into $out"
