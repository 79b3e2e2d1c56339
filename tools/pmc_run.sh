#!/bin/bash
# usage: tools/pmc_run.sh <outdir> [B N mode] ; runs separate rocprofv3 --pmc passes (one counter group per run)
# mode: invert (default; the seven-step sweep of gpsr_invert on B matrices of order N) | eval (one whole fit evaluation, n = N)
set -u
OUT=${1:-gpurun_out/pmc}
B=${2:-633}; N=${3:-784}; MODE=${4:-invert}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # name, counters
  timeout -s KILL 120 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -o $1 -- python3 tools/pmc_invert.py $B $N 1 $MODE > $OUT/$1.log 2>&1 || { echo "pass $1 failed"; tail -3 $OUT/$1.log; return 1; }
}
run fetch "FETCH_SIZE" && run write "WRITE_SIZE" && run l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" && \
run sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64" && \
run grbm "GRBM_GUI_ACTIVE" && echo "all passes ok"
ls -R $OUT | head -40
