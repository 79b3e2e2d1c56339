#!/bin/bash
# One GPU-box session: tests, both bench modes, a kernel trace of the chained mode.  Every GPU step has its own limit
# and the chain stops at the first failure.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread -s > gpurun_out/t2.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/t2.log; tail -3 gpurun_out/t2.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python -u bench.py --steps 3 --warmup 1 > gpurun_out/bench_indep.json 2> gpurun_out/bench_indep.err || { echo bench_indep failed; tail -5 gpurun_out/bench_indep.err; exit 1; }
cat gpurun_out/bench_indep.json | cut -c1-600
timeout -k 10 400 python -u bench.py --warm-start chain_per_image --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_chain.json 2> gpurun_out/bench_chain.err || { echo bench_chain failed; tail -5 gpurun_out/bench_chain.err; exit 1; }
cat gpurun_out/bench_chain.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_chain -o chain -- python3 $GRAFT_REPO_ROOT/bench.py --warm-start chain_per_image --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_chain.log 2>&1 || { echo rocprof failed; tail -5 $GRAFT_REPO_ROOT/gpurun_out/prof_chain.log; exit 1; }
ls $GRAFT_REPO_ROOT/gpurun_out/prof_chain | head
