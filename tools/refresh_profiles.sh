#!/bin/bash
# One GPU session that regenerates everything under profiles/ for the current build (run through gpurun; copies the
# results into gpurun_out/profiles_new/, to be moved to profiles/ and committed from the build container).
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/profiles_new
mkdir -p "$OUT"
cd "$REPO"
echo "== bench (default command)"; timeout -k 10 400 python bench.py > "$OUT/r01_bench.json" 2> "$OUT/bench.err" || exit 1
echo "== clock / power during 60 steps"; tools/clock_sampler.sh "$OUT/clocks_f32.log" -- timeout -k 10 200 python bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-alt-math > "$OUT/bench_60.json" 2>/dev/null
tools/clock_sampler.sh "$OUT/clocks_bf16x3.log" -- timeout -k 10 200 python bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-alt-math --math bf16x3 > "$OUT/bench_60_bf16x3.json" 2>/dev/null
{ echo "bench.py --steps 60 --warmup 3, sysfs freq1_input / power1_input of the loaded card (tools/clock_sampler.sh)";
  for m in f32 bf16x3; do f=$OUT/bench_60.json; [ $m = bf16x3 ] && f=$OUT/bench_60_bf16x3.json;
    echo "math $m: $(python3 -c "import json;d=json.load(open('$f'));print(round(d['value'],1),'samples/s',round(d['ms_per_step'],2),'ms/step')")  $(python3 tools/clock_summary.py $OUT/clocks_$m.log)"; done; } > "$OUT/r01_clock_power.log"
echo "== config bench"; { for m in f32 bf16x3 bf16x3_fast; do echo "NPM_MATH=$m"; NPM_MATH=$m timeout -k 10 300 python tools/config_bench.py --kernels; done; } > "$OUT/r01_config_bench.log" 2>&1
echo "== gemm shapes"; { for t in 10=0 10=2 10=1; do timeout -k 10 200 python tools/gemm_bench.py --tune $t; done; } > "$OUT/r01_gemm_shapes.log" 2>&1
echo "== row kernels"; timeout -k 10 200 python tools/rowops_bench.py > "$OUT/r01_rowops.log" 2>&1
echo "== math error"; timeout -k 10 100 python tools/math_bias.py > "$OUT/r01_math_error.log" 2>&1
echo "== rocprofv3 kernel trace"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-alt-math > "$OUT/r01_bench_under_rocprof.json" 2> "$OUT/prof.err" || exit 1
cd "$REPO"
python3 profiles/summarize_rocprof.py "$OUT"/prof/*/*_kernel_trace.csv --steps 5 --warmup 2 > "$OUT/r01_bench_kernel_trace.md"
cp "$OUT"/prof/*/*_kernel_stats.csv "$OUT/r01_bench_kernel_stats.csv"
echo "== PMC traffic (two passes)"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_$c" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-alt-math > "$OUT/pmc_$c.json" 2> "$OUT/pmc_$c.err" || exit 1
done
cd "$REPO"
python3 profiles/summarize_pmc.py "$OUT"/pmc_FETCH_SIZE/*/*_counter_collection.csv "$OUT"/pmc_WRITE_SIZE/*/*_counter_collection.csv > "$OUT/pmc_traffic.json"
rm -rf "$OUT"/prof "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE
echo "== done"; ls "$OUT"
