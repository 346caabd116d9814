#!/bin/bash
# first GPU pass of round 3: the two-level tests, the wide-tree tests, then the harness-geometry scenarios
set -o pipefail
o=gpurun_out; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_two_level.py tests/test_gpu_wide_and_shards.py -x -q -m gpu > $o/r3a_pytest.log 2>&1
rc=$?; tail -15 $o/r3a_pytest.log
[ $rc -eq 0 ] || exit $rc
tools/gpu_scen.sh r3a harness harness03 l4096
