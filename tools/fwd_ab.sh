# forward / backward of the fused attention core, warm (back-to-back launches), per head size and mode; $1 = extra --tune
T=${1:-}
for d in 128 64 32 16; do
  h=$((1024/d))
  python tools/attn_bench.py --h $h --d $d --save-scores --warm 40 --reps 20 ${T:+--tune $T} 2>&1 | grep "^B \|warm"
done
python tools/attn_bench.py --mask causal --save-scores --warm 40 --reps 20 ${T:+--tune $T} 2>&1 | grep "^B \|warm"
python tools/attn_bench.py --warm 40 --reps 20 ${T:+--tune $T} 2>&1 | grep "^B \|warm"
