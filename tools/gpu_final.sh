#!/bin/bash
# bench line + rehearsals of the N > 1 paths on one GPU + smoke.  Usage: tools/gpu_final.sh <tag>
set -o pipefail
tag=${1:-x}
o=gpurun_out; mkdir -p $o
python -c "import __graft_entry__ as g; g.smoke()" > $o/${tag}_smoke.log 2>&1 || { tail -5 $o/${tag}_smoke.log; exit 1; }
tail -1 $o/${tag}_smoke.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $o/${tag}_bench.json 2> $o/${tag}_bench.err || { echo "bench failed"; tail -5 $o/${tag}_bench.err; exit 1; }
python - <<PY
import json
d=json.loads([l for l in open("$o/${tag}_bench.json") if l.startswith("{")][-1])
print("bench", round(d["value"]/1e6,1), "M reads/s", d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_measured"), d.get("cpu_baseline",{}).get("value"), d.get("want_hits_reads_per_s"), d.get("host_buffers_reads_per_s"))
PY
PFQ_BENCH_SAME_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --reads-per-step 4194304 > $o/${tag}_rehearsal_n2.json 2> $o/${tag}_rehearsal_n2.err || { echo "rehearsal n2 failed"; tail -5 $o/${tag}_rehearsal_n2.err; exit 1; }
PFQ_BENCH_SAME_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --subtree-depth 1 --leaves 2048 --reads-per-step 4194304 --steps 5 --warmup 2 > $o/${tag}_rehearsal_subtree_n2.json 2> $o/${tag}_rehearsal_subtree_n2.err || { echo "subtree rehearsal failed"; tail -5 $o/${tag}_rehearsal_subtree_n2.err; exit 1; }
PFQ_BENCH_FORCE_PG=1 timeout -k 10 300 python bench.py --gpus 1 --steps 5 --warmup 2 --cpu-seconds 0 > $o/${tag}_rccl_one_rank.json 2> $o/${tag}_rccl_one_rank.err || { echo "one-rank RCCL run failed"; tail -5 $o/${tag}_rccl_one_rank.err; exit 1; }
echo "rehearsals ok"
