#!/bin/bash
set -o pipefail
o=gpurun_out; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_two_level.py -x -q -m gpu > $o/r3c_pytest.log 2>&1
rc=$?; tail -5 $o/r3c_pytest.log
[ $rc -eq 0 ] || exit $rc
tools/gpu_scen.sh r3c harness harness03 l4096 t03 t1
