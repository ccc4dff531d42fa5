#!/bin/bash
# usage (on the GPU box, from the repo root): bash scripts/collect_profiles.sh <round-tag> [part ...]
# Regenerates the source data of profiles/<round-tag>/ (written to gpurun_out/<round-tag>/); parts (default: all):
#   bench     bench.py unprofiled: default K/W and the driver's --steps 20 --warmup 5 -> bench_default.json, bench_steps20.json
#   stats     bench.py under rocprofv3 --kernel-trace --stats -> bench_under_rocprof.json, bench_kernel_stats.csv,
#             bench_kernel_trace_xm_kernels.csv, timeline.txt; the same for --dtype c128 -> *_c128.*
#   pmc       separate --pmc passes (one rocprofv3 run per counter group, each with --kernel-trace only -- never with the
#             sys / runtime / hip / hsa / memory-copy trace domains): main kernel of the speculative schedule
#             (k_zf2p mode 13), guess stage (coarse spectra + refine), classic pre-pass -> pmc_main_kernel.txt,
#             pmc_guess_kernel.txt, pmc_prepass_kernel.txt; complex128 main kernel at 65,536 voxels -> pmc_c128_main.txt
#             (scripts/pmc_json.py turns the two main-kernel files into the JSON bench.py reads for roofline.traffic)
#   labs      the small timing scripts: guess stage, heterogeneous family, host path, C1 tolerance, configs, FFT sweep,
#             c128 kernel modes, 6-rank shared-GPU rehearsal of the host budget; round 4: the device search against the
#             host engine (check_device_search.py), zero fill + apodisation in one launch, the search engines and the
#             searches-in-flight sweep on one box, the `configs` record alone
set -e -o pipefail
tag=${1:-r04}
shift || true
parts=${*:-bench stats pmc labs}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
has() { [[ " $parts " == *" $1 "* ]]; }
if has bench; then
  python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
  python3 bench.py --steps 20 --warmup 5 > $out/bench_steps20.json 2> $out/bench_steps20.err
  python3 bench.py --dtype c128 --steps 40 --warmup 5 --no-cpu-baseline --no-footnotes > $out/bench_c128.json 2> $out/bench_c128.err
  echo "bench done"
fi
if has stats; then
  for dt in c64 c128; do
    sfx=""; [ $dt = c128 ] && sfx="_c128"
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --dtype $dt --no-cpu-baseline --no-footnotes \
        2> $out/bench_under_rocprof$sfx.err | grep '^{' > $out/bench_under_rocprof$sfx.json
    cp "$(find $out/stats -name '*kernel_stats.csv' | head -1)" $out/bench_kernel_stats$sfx.csv
    t=$(find $out/stats -name '*kernel_trace.csv' | head -1)
    { head -1 "$t"; grep -E '"(void )?k_' "$t" | tail -700 || true; } > $out/bench_kernel_trace_xm_kernels$sfx.csv
    python3 scripts/trace_timeline.py $out/stats 48 > $out/timeline$sfx.txt
    rm -rf $out/stats
  done
  echo "stats done"
fi
if has pmc; then
  GROUPS_MAIN=("FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"
        "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
        "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum")
  GROUPS_SMALL=("FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU")
  rm -rf gpurun_out/pmc_${tag}*
  bash scripts/pmc.sh ${tag}main all "${GROUPS_MAIN[@]}" > $out/pmc_main_kernel.txt
  rm -rf gpurun_out/pmc_${tag}main_*
  bash scripts/pmc.sh ${tag}guess coarse "${GROUPS_SMALL[@]}" > $out/pmc_guess_kernel.txt
  rm -rf gpurun_out/pmc_${tag}guess_*
  bash scripts/pmc.sh ${tag}pre pre "${GROUPS_SMALL[@]}" > $out/pmc_prepass_kernel.txt
  rm -rf gpurun_out/pmc_${tag}pre_*
  DTYPE=c128 VARIANT_KEY=1 bash scripts/pmc.sh ${tag}c128 all "${GROUPS_MAIN[@]}" > $out/pmc_c128_main.txt
  rm -rf gpurun_out/pmc_${tag}c128_*
  echo "counters done"
fi
if has labs; then
  python3 scripts/time_guess_stage.py > $out/guess_stage.txt 2>/dev/null
  NSETS=12 python3 scripts/time_hetero.py > $out/hetero_steps.txt 2>/dev/null
  { echo "== heterogeneous family, 16 datasets per call (13 of 16 searches need the reference-route polish): ms per dataset";
    echo "-- default: polish in worker processes"; NSETS=16 python3 scripts/time_hetero.py 2>/dev/null | grep "^rep";
    echo "-- XM_POLISH_THREADS=2: polish on helper threads of the launch process"; XM_POLISH_THREADS=2 NSETS=16 python3 scripts/time_hetero.py 2>/dev/null | grep "^rep";
    echo "-- XMRIS_AMD_POLISH=native: round 3's native polish"; XMRIS_AMD_POLISH=native NSETS=16 python3 scripts/time_hetero.py 2>/dev/null | grep "^rep";
  } > $out/hetero_polish.txt
  python3 scripts/time_accessor_host_path.py > $out/host_path.txt 2>/dev/null
  python3 tests/tool_c1_tolerance.py > $out/c1_tolerance.txt 2>/dev/null
  python3 scripts/check_device_search.py > $out/device_search.txt 2>/dev/null || true   # (two degenerate slices diverge: exit 1)
  python3 scripts/time_zf_apod.py > $out/zf_apod.txt 2>/dev/null
  bash scripts/ab_search_engines.sh > $out/search_engines.txt 2>/dev/null; rm -f gpurun_out/ab_*.json gpurun_out/ab.err
  bash scripts/sweep_search_workers.sh > $out/search_workers.txt 2>/dev/null; rm -f gpurun_out/sw_*.json gpurun_out/sw.err
  python3 bench.py --only-configs --no-cpu-baseline > $out/configs.json 2>/dev/null
  python3 scripts/time_configs.py > $out/time_configs.txt 2>/dev/null || true
  python3 scripts/time_fft_sweep.py > $out/fft_sweep.txt 2>/dev/null || true
  python3 scripts/time_c128_modes.py > $out/c128_modes.txt 2>/dev/null || true
  { echo "== 6 ranks sharing ONE GPU (the box allows six processes on it), 16-CPU quota, 8192 voxels per rank, --steps 20 --warmup 5";
    python3 bench.py --gpus 6 --share-gpu --dist-backend gloo --voxels 8192 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -E "^\[rank|^\{" | cut -c1-420;
    echo "== the same with the team budget of EIGHT ranks on this quota (XM_SOLVER_THREADS=8: 16 CPUs - 8 ranks; 4 threads per search in flight)";
    XM_SOLVER_THREADS=8 python3 bench.py --gpus 6 --share-gpu --dist-backend gloo --voxels 8192 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -E "^\[rank|^\{" | cut -c1-420;
  } > $out/rehearsal_6ranks.txt
  echo "labs done"
fi
echo "collected into $out"
