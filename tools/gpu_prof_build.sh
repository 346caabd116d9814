#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o/r3p_trace -o b -- python3 tools/build_bench.py --genomes 512 > $o/r3p_build.log 2>&1 || { tail -5 $o/r3p_build.log; exit 1; }
find $o/r3p_trace -name "*kernel_stats.csv" -exec cp {} $o/r3p_build_kernel_stats.csv \;
rm -rf $o/r3p_trace
tail -2 $o/r3p_build.log; head -8 $o/r3p_build_kernel_stats.csv | cut -c1-220
