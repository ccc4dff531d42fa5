#!/bin/bash
# usage (GPU box): bash scripts/ab_plans.sh -- same-box A/B of builds of the library that differ in their 3 * 2^k plan rows
# (xm_plans.h), run through XMRIS_AMD_LIB: the shipped build against xmris_amd/libxmris_hip_alt{1,2}.so when they exist.
# The builds measured in profiles/r04/ab_plans.txt came from a temporary -DXM_ALT_PLANS switch in xm_plans.h (rows listed in
# that file's header); the outcome -- only 1536 gains, and only for the plain transform -- is `Plan1536Wide` there.
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
for v in shipped alt1 alt2; do
  lib=$here/xmris_amd/libxmris_hip.so
  [ $v != shipped ] && lib=$here/xmris_amd/libxmris_hip_$v.so
  [ -f $lib ] || continue
  echo "== $v"
  XMRIS_AMD_LIB=$lib python3 -c "
import sys, torch; sys.path.insert(0, '$here')
from xmris_amd import device as dev
for n in (768, 1536, 3072):
    x = torch.view_as_complex(torch.randn(64, n, 2, device='cuda', dtype=torch.float64))
    for xx in (x, x.to(torch.complex64)):
        y = dev.fft(xx, -1, shift_out=True); r = torch.fft.fftshift(torch.fft.fft(x, norm='ortho'), dim=-1)
        print('  check n=%d %s max err %.2e' % (n, xx.dtype, (y.to(torch.complex128) - r).abs().max().item()))
"
  XMRIS_AMD_LIB=$lib python3 $here/scripts/time_fft_sweep.py 768 1536 3072
  XMRIS_AMD_LIB=$lib python3 $here/scripts/time_configs.py 2>/dev/null | grep "^C5"
done
