#!/bin/bash
set -o pipefail
o=gpurun_out; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_two_level.py tests/test_gpu_wide_and_shards.py tests/test_cli.py -x -q -m gpu > $o/r3o_pytest.log 2>&1
rc=$?; tail -5 $o/r3o_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "greedy or load_save or prune" > $o/r3o_pytest2.log 2>&1
rc=$?; tail -5 $o/r3o_pytest2.log
[ $rc -eq 0 ] || exit $rc
tools/gpu_scen.sh r3o harness harness03 l4096
timeout -k 10 300 python tools/build_bench.py > $o/r3o_build_bench.log 2>&1; tail -5 $o/r3o_build_bench.log
