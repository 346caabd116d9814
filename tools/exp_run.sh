source tools/exp2.sh
run base PFQ_BIN_DEBUG=0
run nostore PFQ_BIN_DEBUG=1
run nobin PFQ_BIN_DEBUG=2
