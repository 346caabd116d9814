source tools/exp2.sh
BARGS="--leaves 10010 --genome-len 5000 --k 20 --nbits 11981322 --hashes 17 --read-len 100 --threshold 1.0" run harness10010_t10
BARGS="" run base
