#!/bin/bash
# Round-3 measurement batch: kernel stats + PMC of the default bench, the scenario table, the harness geometry profiles, the
# CLI end-to-end rates, the build rate, one config-5 shard.  Everything lands under gpurun_out/r03_*.
set -o pipefail
export TMPDIR=/tmp
o=gpurun_out; mkdir -p $o
step() { echo "== $*"; }
# PART=2: everything but the default bench's passes and the scenario table (tools/gpu_refresh.sh takes those)
if [ "${PART:-all}" != 2 ]; then
step "default bench: kernel trace + PMC"
tools/gpu_profile.sh r03_bench > $o/r03_bench_profile.log 2>&1 || { tail -5 $o/r03_bench_profile.log; exit 1; }
python3 tools/pmc_traffic.py $o/r03_bench_tcc 8388608 1024 1.0 $o/r03_pmc_traffic.json > $o/r03_pmc_traffic.log 2>&1 || { tail -3 $o/r03_pmc_traffic.log; }
rm -rf $o/r03_bench_tcc
cp $o/r03_pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $o/r03_bench.json 2> $o/r03_bench.err || { tail -5 $o/r03_bench.err; exit 1; }
fi
step "harness geometry: kernel trace + PMC (theta 0.3, theta 1)"
H="--leaves 10010 --nbits 11981322 --hashes 17 --k 20 --read-len 100"
PFQ_BENCH_PARITY_READS=500 tools/gpu_profile.sh r03_harness03 $H --threshold 0.3 > $o/r03_harness03_profile.log 2>&1 || { tail -5 $o/r03_harness03_profile.log; exit 1; }
rm -rf $o/r03_harness03_tcc
tools/gpu_profile.sh r03_harness1 $H > $o/r03_harness1_profile.log 2>&1 || { tail -5 $o/r03_harness1_profile.log; exit 1; }
rm -rf $o/r03_harness1_tcc
step "theta 0.3 with 1 % read errors (config 3): kernel trace + PMC"
PFQ_BENCH_READ_ERRORS=0.01 tools/gpu_profile.sh r03_theta03_errors --threshold 0.3 > $o/r03_theta03_errors_profile.log 2>&1 || { tail -5 $o/r03_theta03_errors_profile.log; exit 1; }
rm -rf $o/r03_theta03_errors_tcc
[ "${PART:-all}" = 2 ] || { step "scenarios"
tools/gpu_scen.sh r03s t1 t1e t03 t03e t07e fam4 fam8 fam8t03 fam8t03e fam8t07e fam4t03 l64 l2048 l4096 long1k harness harness03 > $o/r03_scenarios.txt 2>&1 || { tail -5 $o/r03_scenarios.txt; exit 1; }
cat $o/r03_scenarios.txt; }
step "config 5: one shard"
timeout -k 10 280 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --subtree-depth 3 --subtree-index 5 --leaves 16384 > $o/r03_config5_one_shard.json 2> $o/r03_config5_one_shard.err || { tail -3 $o/r03_config5_one_shard.err; exit 1; }
step "build"
python tools/build_bench.py --genomes 1024 > $o/r03_build_1024.json 2>/dev/null; tail -1 $o/r03_build_1024.json | cut -c1-300
step "CLI"
timeout -k 10 400 python tools/harness_bench.py > $o/r03_harness_bench.log 2>&1 || { tail -5 $o/r03_harness_bench.log; exit 1; }
timeout -k 10 400 python tools/cli_bench.py --reads 32000000 --threads 16 --threshold 0.3 > $o/r03_cli_bench_32M_reads.log 2>&1 || { tail -5 $o/r03_cli_bench_32M_reads.log; exit 1; }
echo done
