#!/bin/bash
# rocprofv3 passes of bench.py (program directly after `--`): kernel trace + stats, then the TCC and SQ counter passes on
# their own (--kernel-trace only).  Usage: tools/gpu_profile.sh <tag> [bench.py args...]   Output: gpurun_out/<tag>_*
set -o pipefail
tag=$1; shift
export TMPDIR=/tmp
o=gpurun_out
args="--steps 3 --warmup 2 --cpu-seconds 0 $*"
echo "rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_trace -o bench -- python3 bench.py $args" > $o/${tag}_commands.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_trace -o bench -- python3 bench.py $args > $o/${tag}_trace.log 2>&1 || { tail -5 $o/${tag}_trace.log; exit 1; }
echo "rocprofv3 --kernel-trace --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $o/${tag}_tcc -o bench -- python3 bench.py $args" >> $o/${tag}_commands.txt
rocprofv3 --kernel-trace --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $o/${tag}_tcc -o bench -- python3 bench.py $args > $o/${tag}_tcc.log 2>&1 || { tail -5 $o/${tag}_tcc.log; exit 1; }
echo "rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $o/${tag}_sq -o bench -- python3 bench.py $args" >> $o/${tag}_commands.txt
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $o/${tag}_sq -o bench -- python3 bench.py $args > $o/${tag}_sq.log 2>&1 || { tail -5 $o/${tag}_sq.log; exit 1; }
python3 tools/pmc_parse.py $o/${tag}_tcc > $o/${tag}_tcc_summary.txt
python3 tools/pmc_parse.py $o/${tag}_sq > $o/${tag}_sq_summary.txt
find $o/${tag}_trace -name "*kernel_stats.csv" -exec cp {} $o/${tag}_kernel_stats.csv \;
# keep the merged output small: the raw traces stay on the box
rm -rf $o/${tag}_trace $o/${tag}_sq
ls $o | grep ${tag}
