#!/bin/bash
# After a source change: the whole -m gpu suite, kernel stats + TCC + SQ passes of the default bench (profiles/pmc_traffic.json
# from the TCC pass), the bench line.
set -o pipefail
export TMPDIR=/tmp
o=gpurun_out; mkdir -p $o
tools/gpu_suite.sh r03_final || exit 1
tools/gpu_profile.sh r03_bench > $o/r03_bench_profile.log 2>&1 || { tail -5 $o/r03_bench_profile.log; exit 1; }
python3 tools/pmc_traffic.py $o/r03_bench_tcc 8388608 1024 1.0 $o/r03_pmc_traffic.json > $o/r03_pmc_traffic.log 2>&1 || { tail -3 $o/r03_pmc_traffic.log; exit 1; }
rm -rf $o/r03_bench_tcc
cp $o/r03_pmc_traffic.json profiles/pmc_traffic.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $o/r03_bench.json 2> $o/r03_bench.err || { tail -5 $o/r03_bench.err; exit 1; }
python - <<PY
import json
d=json.loads([l for l in open("$o/r03_bench.json") if l.startswith("{")][-1])
print("bench", round(d["value"]/1e6,1), "M reads/s", round(d["ms_per_step"],2), "frac", round(d["roofline"]["frac"],3), "measured", d["roofline"].get("frac_measured"), "traffic", d["roofline"].get("traffic"))
PY
