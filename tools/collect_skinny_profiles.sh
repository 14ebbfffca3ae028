#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 passes over the weight-streaming kernels, one shape per process
# (tools/ubench/wstream_probe one M N K [nows]): --kernel-trace --stats, then separate --pmc passes
# (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum | TCC_REQ_sum: never combined with other trace domains).
# Raw output under gpurun_out/prof_skinny_$TAG/; tools/pmc_csv_summary.py + the stats CSVs become profiles/r3_skinny_*.
#   usage: gpurun --timeout 900 -- 'tools/collect_skinny_profiles.sh r3'
set -u
TAG=${1:-r3}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_skinny_$TAG
mkdir -p "$OUT"
P=$ROOT/tools/ubench/wstream_probe
cd /tmp && export TMPDIR=/tmp
run() {  # name M N K extra
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${name}_trace" -o t -- $P one "$@" > "$OUT/${name}.txt" 2> "$OUT/${name}_trace.err"
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" TCC_REQ_sum; do
    local d=$OUT/${name}_pmc_$(echo $C | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$d" -o pmc -- $P one "$@" > /dev/null 2> "$d.err"
  done
}
# cfg4's per-GPU shape (OPT-13B fc2 at 32 rows): stream-K kernel (workspace) vs first generation (no workspace)
run fc2_m32_wstream 32 5120 20480
run fc2_m32_skinny  32 5120 20480 nows
run fc2_m16_wstream 16 5120 20480
# cfg1 / decode shapes: first-generation kernel (the dispatcher's choice)
run sq4096_m32      32 4096 4096
run sq4096_m4       4 4096 4096
run gateup_m32      32 11008 4096
run down_m32        32 4096 11008
run w2_m64_wstream  64 4096 14336
ls "$OUT" | head -80
