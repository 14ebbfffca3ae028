for nk in 4096x4096:20 5120x5120:13 4096x11008:8 11008x4096:8 14336x4096:6 8192x8192:5 5120x20480:4 20480x5120:4; do
  shp=${nk%%:*}; rot=${nk##*:}
  SH=""; for m in 48 64 96 128 192 256; do SH="$SH,${m}x${shp}"; done; SH=${SH#,}
  for kern in skinny p8h; do
    ASQ_GEMM_KERNEL=$kern python tools/kbench.py --shapes $SH --batch 4 --iters 8 --rotate $rot | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$kern', d['shape'], d['avg_us'])"
  done
done
