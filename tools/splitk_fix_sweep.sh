#!/bin/bash
# Round 5: K splits of the 128 x 128 kernel, reduced inside the launch (ASQ_SPLITK_FIX=1) / slab launch + reduce launch (=0), forced split counts.
# usage: [KERN=p8q|p8h] [MS=..] [NKS=..] [SS=".."] tools/splitk_fix_sweep.sh > gpurun_out/splitk_fix_sweep.txt
cd "$(dirname "$0")/.."
MS=${MS:-128,256,384,512,768}
NKS=${NKS:-4096x4096,11008x4096,4096x11008}
echo "# tools/midsize_sweep.py --ms $MS --nks $NKS, forced ASQ_GEMM_KERNEL=${KERN:-p8q}, cold weights; S = ASQ_KSPLIT, mode = ASQ_SPLITK_FIX"
echo "== S=1"
python tools/midsize_sweep.py --ms $MS --nks $NKS --env ASQ_GEMM_KERNEL=${KERN:-p8q},ASQ_KSPLIT=1
for S in ${SS:-2 3 4 6 8}; do
  for mode in 1 0; do
    echo "== S=$S mode=$mode"
    python tools/midsize_sweep.py --ms $MS --nks $NKS --env ASQ_GEMM_KERNEL=${KERN:-p8q},ASQ_KSPLIT=$S,ASQ_SPLITK_FIX=$mode
  done
done
