#!/bin/bash
# kernel trace of the harness geometry at theta 0.3 (and theta 1)
set -o pipefail
export TMPDIR=/tmp
o=gpurun_out; mkdir -p $o
H="--leaves 10010 --nbits 11981322 --hashes 17 --k 20 --read-len 100 --steps 3 --warmup 2 --cpu-seconds 0"
PFQ_BENCH_PARITY_READS=500 rocprofv3 --kernel-trace --stats --output-format csv -d $o/r3b_h03_trace -o bench -- python3 bench.py $H --threshold 0.3 > $o/r3b_h03.log 2>&1 || { tail -5 $o/r3b_h03.log; exit 1; }
find $o/r3b_h03_trace -name "*kernel_stats.csv" -exec cp {} $o/r3b_h03_kernel_stats.csv \;
rm -rf $o/r3b_h03_trace
grep '^{' $o/r3b_h03.log | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value']/1e6, {k:d.get(k) for k in ('leaf_groups','coarse_cols','coarse_probes','group_reads_last_step')})"
head -12 $o/r3b_h03_kernel_stats.csv | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d $o/r3b_h1_trace -o bench -- python3 bench.py $H > $o/r3b_h1.log 2>&1 || { tail -5 $o/r3b_h1.log; exit 1; }
find $o/r3b_h1_trace -name "*kernel_stats.csv" -exec cp {} $o/r3b_h1_kernel_stats.csv \;
rm -rf $o/r3b_h1_trace
grep '^{' $o/r3b_h1.log | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value']/1e6, {k:d.get(k) for k in ('leaf_groups','coarse_cols','coarse_probes','group_reads_last_step')})"
head -12 $o/r3b_h1_kernel_stats.csv | cut -c1-200
