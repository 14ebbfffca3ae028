#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the default bench.py line, the same command under
# rocprofv3 --kernel-trace --stats, and separate --pmc passes (FETCH_SIZE, WRITE_SIZE, MfmaUtil, MFMA / LDS counts; never
# combined with other trace domains).  Everything lands in gpurun_out/prof_$TAG/ ;
# tools/summarise_profiles.py then writes the tracked summaries under profiles/.
#   usage: gpurun --timeout 900 -- 'tools/collect_profiles.sh r1'
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python "$ROOT/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
# the traced passes run WITHOUT the cfg3 block: the same kernel templates also serve its M = 65536 launches, which would pollute the
# per-kernel averages of the 4096^3 launches the roofline block is about; cfg3 gets its own --stats pass below
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python "$ROOT/bench.py" --no-cpu-baseline --no-cfg3 --no-other-configs --no-plain-compare > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
for C in FETCH_SIZE WRITE_SIZE MfmaUtil SQ_INSTS_VALU_MFMA_I8 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -o pmc -- python "$ROOT/bench.py" --no-cpu-baseline --no-cfg3 --no-other-configs --no-plain-compare --steps 5 --warmup 2 > "$OUT/pmc_$C.json" 2> "$OUT/pmc_$C.err"
done
for F in "" "--fuse-norm --fuse-qkv"; do
  T=cfg3$( [ -n "$F" ] && echo _fused )
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$T" -o cfg3 -- python "$ROOT/bench.py" --workload llama7b_decoder_b32_s2048 $F --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/$T.json" 2> "$OUT/$T.err"
done
# BASELINE configs[0] (4 x 4096 x 4096, fp32): the whole module forward must be ONE kernel per call (gemm_i8_skinny_fq), and the roctx ranges of the C-ABI (ASQ_ROCTX=1)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg1" -o cfg1 -- python "$ROOT/bench.py" --workload cfg1_int8linear_m4 --dtype f32 --no-cpu-baseline --steps 200 --warmup 20 > "$OUT/cfg1.json" 2> "$OUT/cfg1.err"
ASQ_ROCTX=1 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d "$OUT/cfg1_roctx" -o cfg1 -- python "$ROOT/bench.py" --workload cfg1_int8linear_m4 --dtype f32 --no-cpu-baseline --steps 50 --warmup 5 > "$OUT/cfg1_roctx.json" 2> "$OUT/cfg1_roctx.err"
ls -R "$OUT" | head -60
