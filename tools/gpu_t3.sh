#!/bin/bash
set -o pipefail
o=gpurun_out; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_two_level.py -x -q -m gpu > $o/r3g_pytest.log 2>&1
rc=$?; tail -5 $o/r3g_pytest.log
[ $rc -eq 0 ] || exit $rc
tools/gpu_scen.sh r3g t03 t03e t07e harness03 fam8t03
