#!/bin/bash
# CLI tests + end-to-end CLI rates.  Usage: tools/gpu_cli.sh <tag>
set -o pipefail
tag=$1; o=gpurun_out; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_cli.py tests/test_ingest.py -x -q > $o/${tag}_pytest.log 2>&1
rc=$?; tail -4 $o/${tag}_pytest.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python tools/harness_bench.py > $o/${tag}_harness_bench.log 2>&1 || { tail -5 $o/${tag}_harness_bench.log; exit 1; }
cat $o/${tag}_harness_bench.log
timeout -k 10 500 python tools/cli_bench.py --reads 32000000 --threads 16 --threshold 0.3 > $o/${tag}_cli_bench.log 2>&1 || { tail -5 $o/${tag}_cli_bench.log; exit 1; }
cat $o/${tag}_cli_bench.log | cut -c1-700
