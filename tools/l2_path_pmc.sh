#!/bin/bash
# Round 5: how many bytes do the tiled kernels pull through the L2 per launch, and at what rate?  TCC_REQ_sum (x 128 B), TCC_HIT_sum / TCC_MISS_sum,
# separate --pmc passes with --kernel-trace only, over tools/kbench.py launches of 4096^3 (gemm_i8_p16), 1024 / 512 x 4096 x 4096 (gemm_i8_p8q2) and 2048 x 4096 x 4096.
#   usage (GPU box): tools/l2_path_pmc.sh ; python tools/pmc_csv_summary.py gpurun_out/l2pmc/* --match gemm_i8
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/l2pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SH=4096x4096x4096,2048x4096x4096,1024x4096x4096,512x4096x4096,4096x4096x16384
for C in TCC_REQ_sum "TCC_HIT_sum TCC_MISS_sum" TCC_READ_sum; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$T" -o pmc -- python "$ROOT/tools/kbench.py" --shapes $SH --iters 10 > "$OUT/$T.log" 2> "$OUT/$T.err"
done
python "$ROOT/tools/pmc_csv_summary.py" "$OUT"/TCC_* --match gemm_i8 > "$OUT/summary.txt" 2>&1
tail -40 "$OUT/summary.txt"
