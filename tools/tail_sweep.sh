L=autosmoothquant_amd/libasq_hip.so
for c in 2 4 6 8 11 14 16 20; do
  N=$((256*(32+c)))
  for mode in "ASQ_TAIL_MAX=256,ASQ_TAIL_RMAX=200" "ASQ_TAIL_MAX=256,ASQ_TAIL_RMAX=200,ASQ_TAIL=p8h"; do
    echo -n "c=$c $mode: "; python tools/lib_ab.py --a $L --b $L --shapes 2048x${N}x4096 --a-env ASQ_NO_TAIL=1 --b-env $mode --rounds 4 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/.*| A:/A:/'
  done
done
