#!/bin/bash
set -o pipefail
o=gpurun_out; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_two_level.py tests/test_gpu_wide_and_shards.py -x -q -m gpu > $o/r3n_pytest.log 2>&1
rc=$?; tail -5 $o/r3n_pytest.log
[ $rc -eq 0 ] || exit $rc
tools/gpu_scen.sh r3n harness harness03 l4096
