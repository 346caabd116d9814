#!/bin/bash
# One GPU-box session of the round's standard checks; every step logs under gpurun_out/.  Usage: tools/gpu_round.sh <tag>
# (run through gpurun; steps are joined so that a GPU step that fails stops the ones after it)
set -o pipefail
tag=${1:-x}
export TMPDIR=/tmp
o=gpurun_out
mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/${tag}_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a $o/${tag}_pytest.log; tail -4 $o/${tag}_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > $o/${tag}_bench.json 2> $o/${tag}_bench.err || { echo "bench failed"; tail -5 $o/${tag}_bench.err; exit 1; }
PFQ_BENCH_SAME_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --reads-per-step 4194304 > $o/${tag}_rehearsal_n2.json 2> $o/${tag}_rehearsal_n2.err || { echo "rehearsal n2 failed"; tail -5 $o/${tag}_rehearsal_n2.err; exit 1; }
PFQ_BENCH_SAME_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --subtree-depth 1 --leaves 2048 --reads-per-step 4194304 --steps 5 --warmup 2 > $o/${tag}_rehearsal_subtree_n2.json 2> $o/${tag}_rehearsal_subtree_n2.err || { echo "subtree rehearsal failed"; tail -5 $o/${tag}_rehearsal_subtree_n2.err; exit 1; }
echo "bench + rehearsals ok"
