#!/bin/bash
# usage (GPU box): bash scripts/ab_coarse_variants.sh -- k_coarse_mfma built with prefetch depth 1 / 2 and plain / nontemporal
# loads (xmris_amd/libxmris_hip_c<depth><nt>.so; the shipped library is depth 2, nontemporal), standalone on 65,536 x 4096
# complex64 rows, three rounds
here=$(cd "$(dirname "$0")/.." && pwd)
for round in 1 2 3; do
  for v in shipped c10 c11 c20; do
    lib=$here/xmris_amd/libxmris_hip.so
    [ $v != shipped ] && lib=$here/xmris_amd/libxmris_hip_$v.so
    [ -f $lib ] || continue
    echo "$v: $(XMRIS_AMD_LIB=$lib python3 $here/scripts/time_guess_stage.py 2>/dev/null | grep 'coarse spectra, est')"
  done
done
