#!/bin/bash
# the whole -m gpu suite, log under gpurun_out/<tag>_pytest.log.  Usage: tools/gpu_suite.sh <tag> [pytest args]
set -o pipefail
tag=${1:-x}; shift
o=gpurun_out; mkdir -p $o
if [ $# -gt 0 ] && [ -e "$1" ]; then T=""; else T="tests"; fi
timeout -k 10 1100 python -m pytest $T -x -q -m gpu "$@" > $o/${tag}_pytest.log 2>&1
rc=$?; tail -8 $o/${tag}_pytest.log; exit $rc
