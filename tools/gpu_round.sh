#!/bin/bash
# One GPU-box session: tests, both bench modes, kernel traces (rocprofv3 --stats) of both, PMC passes.  Every GPU step
# has its own limit and the chain stops at the first failure.
set -o pipefail
R=${ROUND:-r03}
mkdir -p gpurun_out
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread -s > gpurun_out/t2.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/t2.log; tail -3 gpurun_out/t2.log
[ $rc -eq 0 ] || exit $rc
fi
timeout -k 10 400 python -u bench.py --steps 3 --warmup 1 > gpurun_out/${R}_bench_set14x3.json 2> gpurun_out/bench_indep.err || { echo bench_indep failed; tail -5 gpurun_out/bench_indep.err; exit 1; }
cut -c1-400 gpurun_out/${R}_bench_set14x3.json
timeout -k 10 400 python -u bench.py --warm-start chain_per_image --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${R}_bench_set14x3_chain_per_image.json 2> gpurun_out/bench_chain.err || { echo bench_chain failed; tail -5 gpurun_out/bench_chain.err; exit 1; }
cut -c1-400 gpurun_out/${R}_bench_set14x3_chain_per_image.json
timeout -k 10 400 python -u bench.py --workload 4kx2 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${R}_bench_4kx2.json 2> gpurun_out/bench_4k.err || { echo bench_4k failed; tail -5 gpurun_out/bench_4k.err; exit 1; }
cut -c1-400 gpurun_out/${R}_bench_4kx2.json
for w in bsd100x3 bsd100x2 p34 agpr; do
timeout -k 10 400 python -u bench.py --workload $w --steps 2 --warmup 1 > gpurun_out/${R}_bench_$w.json 2> gpurun_out/bench_$w.err || { echo bench_$w failed; tail -5 gpurun_out/bench_$w.err; exit 1; }
cut -c1-300 gpurun_out/${R}_bench_$w.json; echo
done
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_indep -o indep -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference-check > $ROOT/gpurun_out/${R}_bench_set14x3_under_rocprof.json 2> $ROOT/gpurun_out/prof_indep.log || { echo rocprof indep failed; tail -5 $ROOT/gpurun_out/prof_indep.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_chain -o chain -- python3 $ROOT/bench.py --warm-start chain_per_image --steps 1 --warmup 0 --no-cpu-baseline --no-reference-check > $ROOT/gpurun_out/prof_chain.json 2> $ROOT/gpurun_out/prof_chain.log || { echo rocprof chain failed; tail -5 $ROOT/gpurun_out/prof_chain.log; exit 1; }
cd $ROOT
# keep the per-kernel statistics only: the full kernel traces (190 000 rows for a chained step) would push gpurun_out/ past
# the 64 MiB that are copied back
for d in prof_indep prof_chain; do
  find gpurun_out/$d -name "*kernel_stats.csv" -exec cp {} gpurun_out/${R}_${d}_kernel_stats.csv \;
  rm -rf gpurun_out/$d
done
if [ "${SKIP_PMC:-0}" != "1" ]; then
bash tools/pmc_run.sh gpurun_out/pmc324 10800 324 eval && python3 tools/pmc_summarize.py gpurun_out/pmc324 gpurun_out/${R}_pmc_eval_10800x324.json 10800 324 eval || exit 1
bash tools/pmc_run.sh gpurun_out/pmc324f0 10800 324 fit1 && python3 tools/pmc_summarize.py gpurun_out/pmc324f0 gpurun_out/${R}_pmc_fit1_10800x324.json 10800 324 fit1 || exit 1
GPSR_FUSED=1 bash tools/pmc_run.sh gpurun_out/pmc324f1 10800 324 fit1 && GPSR_FUSED=1 python3 tools/pmc_summarize.py gpurun_out/pmc324f1 gpurun_out/${R}_pmc_fit1_fused_10800x324.json 10800 324 fit1 || exit 1
bash tools/pmc_run.sh gpurun_out/pmc784 633 784 invert && python3 tools/pmc_summarize.py gpurun_out/pmc784 gpurun_out/${R}_pmc_invert_633x784.json 633 784 invert || exit 1
rm -rf gpurun_out/pmc324 gpurun_out/pmc324f0 gpurun_out/pmc324f1 gpurun_out/pmc784
fi
du -sh gpurun_out
echo round-ok
