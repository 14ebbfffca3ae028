#!/bin/bash
# Dev tool: sweep the forced split-K count for one forced kernel.  usage: KERN=p8h SHAPES="a b c" KS="0 1 2 3" tools/ksplit_sweep.sh
KERN=${KERN:-p8}
SHAPES=${SHAPES:-"512x4096x4096 1024x4096x4096 2048x4096x4096"}
KS=${KS:-"0 1 2 3 4 6 8"}
for sh in $SHAPES; do
  for ks in $KS; do
    if [ $ks = 0 ]; then unset ASQ_KSPLIT; else export ASQ_KSPLIT=$ks; fi
    ASQ_GEMM_KERNEL=$KERN python tools/kbench.py --shapes $sh --iters 20 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$KERN','$sh','ks=$ks',d['avg_us'],d['min_us'])"
  done
done
