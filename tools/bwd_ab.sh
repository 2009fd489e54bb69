# backward of the fused attention core at the C4 shape, warm; $1 = extra --tune
T=${1:-}
python tools/attn_bench.py --save-scores --warm 30 --reps 20 ${T:+--tune $T} 2>&1 | grep "^B \|warm"
