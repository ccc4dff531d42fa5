#!/bin/bash
# usage (on the GPU box, from the repo root): bash scripts/collect_profiles.sh <round-tag>
# Regenerates the source data of profiles/<round-tag>/ (written to gpurun_out/<round-tag>/):
#   1. bench.py unprofiled (default K/W and the driver's --steps 20 --warmup 5) -> bench_default.json, bench_steps20.json
#   2. bench.py under rocprofv3 --kernel-trace --stats -> bench_under_rocprof.json, bench_kernel_stats.csv,
#      bench_kernel_trace_xm_kernels.csv
#   3. separate --pmc passes (one rocprofv3 run per counter group, each with --kernel-trace only -- never with the
#      sys / runtime / hip / hsa / memory-copy trace domains) for the main kernel of the
#      speculative schedule (k_zf2p mode 13), the guess kernel and the classic schedule's pre-pass
#      -> pmc_main_kernel.txt, pmc_guess_kernel.txt, pmc_prepass_kernel.txt
#   3b. complex128: c128_modes.txt, pmc_c128_main.txt (k_zf2d, then k_zf2<double>), bench_c128.json, timeline_c128.txt
#   4. the labs behind DESIGN.md section 4: streaming ceilings and kernel variants -> stream_lab.txt, zf2_lab.txt,
#      stream_ceiling.txt
set -e -o pipefail
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_steps20.json 2> $out/bench_steps20.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline --no-footnotes \
    2> $out/bench_under_rocprof.err | grep '^{' > $out/bench_under_rocprof.json
f=$(find $out/stats -name '*kernel_stats.csv' | head -1)
cp "$f" $out/bench_kernel_stats.csv
t=$(find $out/stats -name '*kernel_trace.csv' | head -1)
{ head -1 "$t"; grep -E '"(void )?k_' "$t" | tail -700 || true; } > $out/bench_kernel_trace_xm_kernels.csv
rm -rf $out/stats
# small groups: a pass that asks for more counters than the hardware can collect at once aborts
PMC_GROUPS=("FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"
        "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
        "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum")
rm -rf gpurun_out/pmc_${tag}main_* gpurun_out/pmc_${tag}pre_* gpurun_out/pmc_${tag}guess_*
bash scripts/pmc.sh ${tag}main all "${PMC_GROUPS[@]}" > $out/pmc_main_kernel.txt
echo "main-kernel counters done"
bash scripts/pmc.sh ${tag}guess guess "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU" > $out/pmc_guess_kernel.txt
bash scripts/pmc.sh ${tag}pre pre "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" > $out/pmc_prepass_kernel.txt
rm -rf gpurun_out/pmc_${tag}main_* gpurun_out/pmc_${tag}pre_* gpurun_out/pmc_${tag}guess_*
echo "counters done"
make -C tools -j4 > /dev/null 2>&1 || true
./tools/stream_lab 65536 5 > $out/stream_lab.txt
./tools/zf2_lab 65536 7 > $out/zf2_lab.txt
./tools/stream_ceiling > $out/stream_ceiling.txt
# complex128 (the reference's arithmetic): kernel modes old / new, counters of both main kernels, timeline of the steps
{ echo "== k_zf2d (default)"; python3 scripts/time_c128_modes.py 2>/dev/null; echo "== XM_ZF2D_GEN1=1: k_zf2<double>"; XM_ZF2D_GEN1=1 python3 scripts/time_c128_modes.py 2>/dev/null; } > $out/c128_modes.txt
C128_GROUPS=("FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"
        "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
        "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU")
rm -rf gpurun_out/pmc_${tag}c128*
{ NV=32768 DTYPE=c128 bash scripts/pmc.sh ${tag}c128 rows "${C128_GROUPS[@]}"; rm -rf gpurun_out/pmc_${tag}c128_*;
  XM_ZF2D_GEN1=1 NV=32768 DTYPE=c128 bash scripts/pmc.sh ${tag}c128 rows "${C128_GROUPS[@]}"; } > $out/pmc_c128_main.txt
rm -rf gpurun_out/pmc_${tag}c128*
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/tl_c128 -- python3 bench.py --dtype c128 --voxels 32768 \
    --steps 40 --warmup 5 --no-cpu-baseline --no-footnotes > $out/bench_c128.json 2> $out/bench_c128.err
python3 scripts/trace_timeline.py $out/tl_c128 40 > $out/timeline_c128.txt
rm -rf $out/tl_c128
python3 scripts/time_fill.py > $out/fill_timeline.txt 2>/dev/null || true
python3 scripts/time_fft_sweep.py > $out/fft_sweep.txt 2>/dev/null || true
python3 scripts/time_configs.py > $out/time_configs.txt 2>/dev/null || true
echo "collected into $out"
