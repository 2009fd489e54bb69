# bench.py (headline step only) alternating NPM_TUNE settings: $1 and $2 (e.g. 18=0 18=1), $3 rounds
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do for t in $A $B; do
  echo "NPM_TUNE=$t: $(NPM_TUNE=$t timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-alt-math --no-configs --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value'],1),'samples/s', round(d['ms_per_step'],3),'ms; TN', round(r['by_layout']['sgemm_TN']['avg_ms'],3), 'NN', round(r['by_layout']['sgemm_NN']['avg_ms'],3), 'NT', round(r['by_layout']['sgemm_NT']['avg_ms'],3))")"
done; done
