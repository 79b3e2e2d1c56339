set -o pipefail
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for w in bsd100x3 bsd100x2 p34; do
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$w -o k -- python3 $ROOT/bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-reference-check --no-kernel-breakdown > $ROOT/gpurun_out/r03_bench_${w}_under_rocprof.json 2> $ROOT/gpurun_out/prof_$w.log || exit 1
  find $ROOT/gpurun_out/prof_$w -name "*kernel_stats.csv" -exec cp {} $ROOT/gpurun_out/r03_bench_${w}_kernel_stats.csv \;
  rm -rf $ROOT/gpurun_out/prof_$w
  echo "== $w"; head -4 $ROOT/gpurun_out/r03_bench_${w}_kernel_stats.csv | cut -c1-150
done
