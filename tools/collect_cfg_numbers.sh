#!/bin/bash
# Runs ON THE GPU BOX: one bench.py line per BASELINE config / DESIGN table row -> gpurun_out/cfg_$TAG/*.json
set -u
TAG=${1:-r3}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/cfg_$TAG; mkdir -p $O
cd $ROOT
B="python bench.py --no-cpu-baseline"
$B --workload gemm4096 > $O/gemm4096.json 2> $O/err.txt
$B --workload cfg1_int8linear_m4 --dtype f32 --steps 200 > $O/cfg1_eager.json 2>> $O/err.txt
$B --workload cfg1_int8linear_m4 --dtype f32 --steps 200 --graph > $O/cfg1_graph.json 2>> $O/err.txt
$B --workload llama7b_decode_m32 --steps 200 > $O/decode_m32_eager.json 2>> $O/err.txt
$B --workload llama7b_decode_m32 --steps 200 --graph > $O/decode_m32_graph.json 2>> $O/err.txt
$B --workload opt13b_fc2 --steps 200 > $O/opt_fc2_m256.json 2>> $O/err.txt
$B --workload opt13b_fc2_m32 --steps 200 > $O/opt_fc2_m32_eager.json 2>> $O/err.txt
$B --workload opt13b_fc2_m32 --steps 200 --graph > $O/opt_fc2_m32_graph.json 2>> $O/err.txt
$B --workload mixtral_experts --steps 20 --warmup 5 > $O/mixtral_int8.json 2>> $O/err.txt
$B --workload mixtral_experts --fp8 --steps 20 --warmup 5 > $O/mixtral_fp8.json 2>> $O/err.txt
for w in llama7b_attn_block_b1_s2048 llama7b_attn_block_b1_s128 llama7b_attn_block_b1_s1; do
  $B --workload $w --fuse-norm --fuse-qkv --steps 50 > $O/${w}_fused.json 2>> $O/err.txt
  $B --workload $w --fuse-norm --fuse-qkv --steps 50 --graph > $O/${w}_fused_graph.json 2>> $O/err.txt
done
$B --workload llama7b_layer_linears > $O/layer_linears.json 2>> $O/err.txt
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print(sys.argv[1].split("/")[-1], "value", d.get("value"), d.get("unit"), "ms/step", d.get("ms_per_step"), "tok/s", d.get("tokens_per_s"), "| roofline", r.get("kernel","")[:60], r.get("avg_us"), r.get("frac"), {k:v for k,v in d.items() if k in ("moe",)})
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
tail -5 $O/err.txt
