source tools/exp2.sh
BARGS="" run base
BARGS="--threshold 0.3" run t03_clean
BARGS="--threshold 0.3" run t03_err PFQ_BENCH_READ_ERRORS=0.01
BARGS="--threshold 0.7" run t07_err PFQ_BENCH_READ_ERRORS=0.01
BARGS="--threshold 1.0" run t10_err PFQ_BENCH_READ_ERRORS=0.01
