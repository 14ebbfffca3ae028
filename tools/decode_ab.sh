#!/bin/bash
# Runs ON THE GPU BOX: A/B of the fused prologue (decode / cfg1 shapes) and of the shared activation quantiser.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
OUT=gpurun_out/decode_ab; mkdir -p $OUT
j() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', 'ms_per_step', d['ms_per_step'], 'value', d['value'], 'roofline', d['roofline']['achieved'], d['roofline']['unit'], d['roofline']['avg_us'], 'us')"; }
for F in 1 0; do
  if [ $F = 0 ]; then export ASQ_FUSE_PROLOGUE_BYTES=0; else unset ASQ_FUSE_PROLOGUE_BYTES; fi
  python bench.py --workload llama7b_decode_m32 --no-cpu-baseline --steps 300 --warmup 50 > $OUT/decode_f$F.json 2>/dev/null; j $OUT/decode_f$F.json "decode_m32 fused=$F eager"
  python bench.py --workload llama7b_decode_m32 --no-cpu-baseline --steps 300 --warmup 50 --graph > $OUT/decode_f${F}_graph.json 2>/dev/null; j $OUT/decode_f${F}_graph.json "decode_m32 fused=$F graph"
  python bench.py --workload cfg1_int8linear_m4 --dtype f32 --no-cpu-baseline --steps 500 --warmup 50 > $OUT/cfg1_f$F.json 2>/dev/null; j $OUT/cfg1_f$F.json "cfg1 m4 f32 fused=$F eager"
  python bench.py --workload cfg1_int8linear_m4 --dtype f32 --no-cpu-baseline --steps 500 --warmup 50 --graph > $OUT/cfg1_f${F}_graph.json 2>/dev/null; j $OUT/cfg1_f${F}_graph.json "cfg1 m4 f32 fused=$F graph"
  echo "--- pyoverhead fused=$F"; python tools/pyoverhead.py 2>&1 | grep -E "per call us|bare ctypes|torch.empty us|ops.linear"
done
unset ASQ_FUSE_PROLOGUE_BYTES
for C in 1 0 1 0; do
  ASQ_ACT_CACHE=$C python bench.py --no-cpu-baseline --no-cfg3 > $OUT/attn_c$C.json 2>/dev/null; j $OUT/attn_c$C.json "attn_linears act_cache=$C"
done
