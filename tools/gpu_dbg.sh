#!/bin/bash
o=gpurun_out; mkdir -p $o
for t in "tests/test_gpu_wide_and_shards.py::test_replicas_on_one_device_allreduce_to_single_tree_counts" "tests/test_gpu_wide_and_shards.py::test_replicas_of_a_database_with_stored_counts"; do
  timeout -k 10 300 python -X faulthandler -m pytest "$t" -x -q -m gpu > $o/dbg.log 2>&1; echo "rc=$? $t"; tail -25 $o/dbg.log | grep -v "^$" | head -40
done
which gdb valgrind
