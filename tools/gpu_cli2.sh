#!/bin/bash
set -o pipefail
tag=$1; o=gpurun_out; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_cli.py tests/test_ingest.py -x -q > $o/${tag}_pytest.log 2>&1
rc=$?; tail -4 $o/${tag}_pytest.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python tools/cli_bench.py --reads 32000000 --threads 16 --threshold 0.3 > $o/${tag}_cli_bench.log 2>&1 || { tail -5 $o/${tag}_cli_bench.log; exit 1; }
grep "pos+neg\|counts-only\", \"threads\": 16, \"reads\": 32" $o/${tag}_cli_bench.log | cut -c1-800
timeout -k 10 500 python tools/cli_bench.py --reads 32000000 --threads 16 --threshold 0.3 --block 1000 > $o/${tag}_cli_bench_b1000.log 2>&1 || { tail -5 $o/${tag}_cli_bench_b1000.log; exit 1; }
grep "pos+neg" $o/${tag}_cli_bench_b1000.log | cut -c1-800
