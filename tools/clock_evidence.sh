#!/bin/bash
# Runs ON THE GPU BOX (gpurun): clock / power evidence for the dominant kernel.
#   1. SMI static limits + a background SMI power/clock log while the probe loops (>= 1 s per arm)
#   2. tools/ubench/clock_probe: in-kernel shader clock (s_memtime / s_memrealtime), per-block cycle budget,
#      MFMA-only ceiling on the bench's own operand statistics
#   3. rocprofv3 --pmc passes (never combined with other trace domains) with the raw cycle counters
# Output: gpurun_out/clock_<tag>/ ; summaries are copied to profiles/ by hand (tools/README.md).
#   usage: gpurun --timeout 900 -- 'tools/clock_evidence.sh r2'
set -u
TAG=${1:-r2}
SECS=${2:-2.0}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/clock_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P=$ROOT/tools/ubench/clock_probe
{
  echo "## rocm-smi static"; timeout 60 rocm-smi --showmaxpower --showpower --showclocks --showperflevel 2>&1 | head -60
  echo "## amd-smi static limits"; timeout 60 amd-smi static --limit --clock 2>&1 | head -80
} > "$OUT/smi_static.txt"
# background SMI logger (python startup makes each sample ~0.3-0.5 s; the probe's own sysfs sampler runs at 20 ms)
( while true; do date +%s.%N; timeout 20 amd-smi metric --power --clock --json 2>/dev/null | tr -d '\n' | head -c 6000; echo; sleep 0.2; done ) > "$OUT/smi_log.txt" 2>&1 &
SMI_PID=$!
"$P" "$SECS" all > "$OUT/clock_probe.txt" 2>&1
kill $SMI_PID 2>/dev/null
wait $SMI_PID 2>/dev/null
tail -60 "$OUT/clock_probe.txt"
# PMC passes: SQ has 8 slots, GRBM 2
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_CYCLES \
  --output-format csv -d "$OUT/pmc_cycles" -o pmc -- "$P" 1 pmc > "$OUT/pmc_cycles.log" 2>&1
rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d "$OUT/pmc_mfmautil" -o pmc -- "$P" 1 pmc > "$OUT/pmc_mfmautil.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS \
  --output-format csv -d "$OUT/pmc_insts" -o pmc -- "$P" 1 pmc > "$OUT/pmc_insts.log" 2>&1
python "$ROOT/tools/pmc_csv_summary.py" "$OUT/pmc_cycles" "$OUT/pmc_mfmautil" "$OUT/pmc_insts" --period 2 > "$OUT/pmc_summary.json" 2> "$OUT/pmc_summary.err"
cat "$OUT/pmc_summary.json" | head -80
python - "$OUT/smi_log.txt" <<'PY' > "$OUT/smi_summary.txt" 2>&1
import json, sys
lines = open(sys.argv[1]).read().splitlines()
n = 0
for i in range(0, len(lines) - 1, 2):
    try:
        t = float(lines[i]); d = json.loads(lines[i + 1])
    except Exception:
        continue
    g = d[0] if isinstance(d, list) else d.get("gpu_data", [d])[0]
    pw, ck = g.get("power", {}), g.get("clock", {})
    gfx = {k: v for k, v in ck.items() if k.startswith("gfx")}
    clks = [v.get("clk", {}).get("value") if isinstance(v.get("clk"), dict) else v.get("clk") for v in gfx.values()]
    print(f"{t:.2f} socket_power={pw.get('socket_power')} gfx_clk(min/max of {len(clks)})={min([c for c in clks if isinstance(c,(int,float))], default=None)}/{max([c for c in clks if isinstance(c,(int,float))], default=None)}")
    n += 1
print("samples", n)
PY
tail -25 "$OUT/smi_summary.txt"
ls "$OUT"
