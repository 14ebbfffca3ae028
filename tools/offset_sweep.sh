#!/bin/bash
# Dev tool (round 4): sweep of the two offset constants on the product path (asq_linear_w8a8_off on the library's own images), 4096^3, bench statistics.
# usage: tools/offset_sweep.sh > profiles/r4_offset_constants_sweep.txt
for cx in 2 3 4 6; do
  echo "== ASQ_OFF_CW=64 cx=$cx"; ASQ_OFF_CW=64 python tools/offset_ab.py --cx $cx --arms base,off_exact --rounds 8 | grep -v amdgpu.ids | tail -2
done
for cw in 32 48 96 127; do
  echo "== ASQ_OFF_CW=$cw cx=3"; ASQ_OFF_CW=$cw python tools/offset_ab.py --cx 3 --arms base,off_exact --rounds 8 | grep -v amdgpu.ids | tail -2
done
