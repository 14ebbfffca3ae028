#!/bin/bash
# Round 5: the dispatcher's choice against the two 128-row kernels forced, at model shapes outside LLaMA-7B's three (OPT-13B, Mixtral, fused QKV), cold weights.
cd "$(dirname "$0")/.."
NKS=${NKS:-5120x20480,12288x4096,5120x5120,14336x4096,4096x14336,20480x5120,11008x4096}
MS=${MS:-192,256,384,512,640}
echo "== default"; python tools/midsize_sweep.py --ms $MS --nks $NKS
echo "== forced p8q"; python tools/midsize_sweep.py --ms $MS --nks $NKS --env ASQ_GEMM_KERNEL=p8q
echo "== forced p8h"; python tools/midsize_sweep.py --ms $MS --nks $NKS --env ASQ_GEMM_KERNEL=p8h
