#!/bin/bash
# usage (GPU box): bash scripts/sweep_resident_cap.sh -- XM_RESIDENT_CAP (workgroups per CU of every persistent grid) against
# the staged FFT seam, the coarse-spectra kernel and the fused passes of the configs
for c in 0 1 2 3 4 6; do
  echo "== XM_RESIDENT_CAP=$c"
  XM_RESIDENT_CAP=$c python3 scripts/time_fft_sweep.py 512 1024 1536 2048 4096 8192 2>/dev/null
  XM_RESIDENT_CAP=$c python3 scripts/time_guess_stage.py 2>/dev/null | grep -E "coarse spectra, est|complex128 rows|refine"
  XM_RESIDENT_CAP=$c python3 scripts/time_configs.py 2>/dev/null | grep -E "^C2 |^C3 |^C5 |no zero fill"
done
