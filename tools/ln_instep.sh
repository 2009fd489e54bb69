# LayerNorm kernels inside the headline step for NPM_TUNE settings $@ (e.g. 19=0 19=1 19=2), alternating, 2 rounds
for i in 1 2; do for t in "$@"; do
  echo "NPM_TUNE=$t: $(NPM_TUNE=$t timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-alt-math --no-configs --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['roofline']['hbm_kernels']; print(round(d['ms_per_step'],3),'ms/step;', {k: (round(v['avg_ms'],4), round(v.get('frac_of_8TBps', v.get('frac', 0)),3)) for k, v in h.items()})")"
done; done
