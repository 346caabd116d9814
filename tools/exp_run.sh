source tools/exp2.sh
BARGS="" run base
BARGS="--leaves 4096" run leaves4096
BARGS="--subtree-depth 3 --subtree-index 5 --leaves 16384" run config5_shard
BARGS="" run fam8 PFQ_BENCH_FAMILY=8
