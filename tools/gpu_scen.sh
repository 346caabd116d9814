#!/bin/bash
# Scenario lines of bench.py (kernel times per step).  Usage: tools/gpu_scen.sh <tag> <scenario>...   scenarios: t1 t1e t03 t03e t07e fam4 fam8 fam8t03 harness harness03
set -o pipefail
tag=$1; shift
o=gpurun_out; mkdir -p $o
run() {  # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 280 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 "$@" > $o/${tag}_$name.json 2> $o/${tag}_$name.err || { echo "$name FAILED"; tail -3 $o/${tag}_$name.err; return 1; }
  python - <<PY
import json
d=json.loads([l for l in open("$o/${tag}_$name.json") if l.startswith("{")][-1])
print("%-10s %7.1f M reads/s  " % ("$name", d["value"]/1e6), {k: round(v,2) for k,v in d["kernel_ms_per_step"].items()}, "hits", d.get("hits_last_step"), "cand", d.get("candidates_last_step"), "fallback", d.get("fallback_pairs"))
PY
}
for s in "$@"; do
case $s in
t1) run t1 X=1 -- ;;
neg) run neg PFQ_BENCH_ALL_NEGATIVE=1 -- ;;
t1e) run t1e PFQ_BENCH_READ_ERRORS=0.01 -- ;;
t03) run t03 X=1 -- --threshold 0.3 ;;
t03e) run t03e PFQ_BENCH_READ_ERRORS=0.01 -- --threshold 0.3 ;;
t07e) run t07e PFQ_BENCH_READ_ERRORS=0.01 -- --threshold 0.7 ;;
fam4) run fam4 PFQ_BENCH_FAMILY=4 -- ;;
fam8) run fam8 PFQ_BENCH_FAMILY=8 -- ;;
fam8d0) run fam8d0 PFQ_BENCH_FAMILY=8 PFQ_BENCH_DIVERGENCE=0 -- ;;
t1blk) run t1blk PFQ_BLOCK=1 -- ;;
fam8t03) run fam8t03 PFQ_BENCH_FAMILY=8 -- --threshold 0.3 ;;
fam8t03e) run fam8t03e PFQ_BENCH_FAMILY=8 PFQ_BENCH_READ_ERRORS=0.01 -- --threshold 0.3 ;;
fam8t07e) run fam8t07e PFQ_BENCH_FAMILY=8 PFQ_BENCH_READ_ERRORS=0.01 -- --threshold 0.7 ;;
fam4t03) run fam4t03 PFQ_BENCH_FAMILY=4 -- --threshold 0.3 ;;
l64) run l64 X=1 -- --leaves 64 ;;
l2048) run l2048 X=1 -- --leaves 2048 ;;
l4096) run l4096 X=1 -- --leaves 4096 ;;
long1k) run long1k X=1 -- --read-len 1000 --reads-per-step 1048576 --threshold 0.5 ;;
harness) run harness X=1 -- --leaves 10010 --nbits 11981322 --hashes 17 --k 20 --read-len 100 ;;
harness03) run harness03 PFQ_BENCH_PARITY_READS=500 -- --leaves 10010 --nbits 11981322 --hashes 17 --k 20 --read-len 100 --threshold 0.3 ;;
esac || exit 1
done
