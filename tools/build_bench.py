"""`build` on the device: greedy insertion of N synthetic 50 kb genomes into an SBT with the
BASELINE filter geometry (71 887 936 bits, 10 hashes), timed as a whole; run under `rocprofv3 --kernel-trace --stats`
for the per-kernel view (k_insert_step streams 4 filters and writes 1: 5 x 8.99 MB per launch).

    python tools/build_bench.py [--genomes 1024]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=1024)
    a = ap.parse_args()
    import torch
    from phagefilter_amd import BloomTree, _ffi

    L = _ffi.lib()
    n_g, glen = a.genomes, 50000
    dg = torch.empty(n_g * glen, dtype=torch.uint8, device="cuda")
    _ffi.check(L.pfq_synth_genomes_device(dg.data_ptr(), n_g, glen, 0x5EED0000, None))
    torch.cuda.synchronize()
    genomes = dg.cpu().numpy().reshape(n_g, glen)
    del dg
    # --false-pos-rate 0.001 --largest-genome 5000000 => 71 887 936 bits, 10 hashes (bloom_filter.rs:342-357)
    t0 = time.time()
    gt = BloomTree.new(21, 0.001, 5000000, 0x0123456789ABCDEF, 0xFEDCBA9876543210, expected_genomes=n_g)
    marks = []
    for i in range(n_g):
        gt.insert(genomes[i].tobytes(), f"G{i:05d}")
        if (i + 1) % 128 == 0:
            marks.append(round(time.time() - t0, 3))
    t_ins = time.time() - t0
    info = gt.info()          # waits for the insertions, reads the shape back, renumbers, verifies parent ⊇ child on every edge
    torch.cuda.synchronize()
    wall = time.time() - t0
    depth = []
    counts = gt.get_leaf_counts()
    print(json.dumps({"genomes": n_g, "nodes": info.n_nodes, "leaves": info.n_leaves, "nbits": info.nbits,
                      "num_hashes": info.num_hashes, "superset_verified": info.superset_verified,
                      "build_seconds": round(wall, 3), "genomes_per_s": round(n_g / wall, 1), "insert_calls_seconds": round(t_ins, 3),
                      "seconds_after_every_128_insertions": marks,
                      "filter_bytes": info.n_nodes * ((info.nbits + 63) // 64) * 8}))
    gt.close()


if __name__ == "__main__":
    main()
