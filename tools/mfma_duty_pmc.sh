#!/bin/bash
# Round 6: the cycle counters behind "MfmaUtil" for the dominant kernel -- GRBM_GUI_ACTIVE, SQ_BUSY_CYCLES, SQ_VALU_MFMA_BUSY_CYCLES,
# SQ_INSTS_VALU_MFMA_I8, SQ_WAVE_CYCLES, SQ_WAVES -- one --pmc pass each (--kernel-trace only), over product-path launches of 4096^3 on the bench operands
# (tools/lib_ab_off.py, offset images) and on zeros.  Under counter collection rocprofv3 serialises the dispatches (start / stop packets around each), so the
# chip idles between launches, is not power-limited and clocks near its maximum: the per-dispatch wall time and cycle counts of THESE passes are those of an
# isolated launch, not of the bench's back-to-back train.  profiles/r6_mfma_duty.md puts the two side by side.
#   usage (GPU box): tools/mfma_duty_pmc.sh ; python tools/pmc_csv_summary.py gpurun_out/duty/* --match gemm_i8
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/duty
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
L=$ROOT/autosmoothquant_amd/libasq_hip.so
for Z in "" "--zeros"; do
  for C in GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_WAVE_CYCLES SQ_WAVES MfmaUtil; do
    T=${C}$( [ -n "$Z" ] && echo _zeros )
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$T" -o pmc -- python "$ROOT/tools/lib_ab_off.py" --a $L --b $L --shapes 4096x4096x4096 --rounds 2 --settle 200 $Z > "$OUT/$T.log" 2> "$OUT/$T.err"
  done
done
python "$ROOT/tools/pmc_csv_summary.py" "$OUT"/*/ --match gemm_i8_p16 > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt" | head -80
# the same launches WITHOUT counters: the back-to-back train's per-launch time (for the clock it implies)
python "$ROOT/tools/lib_ab_off.py" --a $L --b $L --shapes 4096x4096x4096 > "$OUT/train.log" 2>&1
python "$ROOT/tools/lib_ab_off.py" --a $L --b $L --shapes 4096x4096x4096 --zeros > "$OUT/train_zeros.log" 2>&1
cat "$OUT/train.log" "$OUT/train_zeros.log" | grep -v amdgpu
