#!/bin/bash
# POS/NEG end to end with different thread splits (the FASTQ is generated once per call of cli_bench.py)
o=gpurun_out; mkdir -p $o
for cfg in "16 16 999" "16 8 4" "12 8 4" "16 12 8" "8 8 4"; do
  set -- $cfg
  echo "== -t $1 formatters $2 writers $3"
  PFQ_CLI_FMT_WORKERS=$2 PFQ_CLI_WRITERS=$3 timeout -k 10 300 python tools/cli_bench.py --reads 32000000 --threads $1 --threshold 0.3 --only posneg 2>/dev/null | grep pos+neg | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  ', d['query_loop'][:75], '|', d['output'][:50], '|', d['cpu'][:60])"
done
