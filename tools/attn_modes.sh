# fused attention core, warm timings: saved scores against the recomputing mode per head size (H D = 1024, B 256, S 512)
for d in 128 64 32 16; do
  h=$((1024/d))
  python tools/attn_bench.py --h $h --d $d --save-scores --warm 30 --reps 15 2>&1 | grep "^B \|warm"
  python tools/attn_bench.py --h $h --d $d --warm 30 --reps 15 2>&1 | grep "^B \|warm"
done
for d in 128 64 32 16; do
  h=$((1024/d))
  python tools/attn_bench.py --h $h --d $d --mask causal --save-scores --warm 30 --reps 15 2>&1 | grep "^B \|warm"
  python tools/attn_bench.py --h $h --d $d --mask causal --warm 30 --reps 15 2>&1 | grep "^B \|warm"
done
