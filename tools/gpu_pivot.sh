#!/bin/bash
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_refpin.py -x -q --timeout 300 --timeout-method thread > gpurun_out/t3.log 2>&1
rc=$?; tail -3 gpurun_out/t3.log; [ $rc -eq 0 ] || exit $rc
for cfg in "633 784 3" "14 784 20" "1 784 20" "10800 336 2"; do
  echo "== $cfg"; timeout -k 10 120 python tools/time_invert.py $cfg || exit 1
done
