import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import test_gpu_full_geometry as T
from hipbuf import DeviceBuffer, synchronize
from oracle import pfq_oracle as orc
from phagefilter_amd import BloomTree, _ffi
L = _ffi.lib()
n_leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d_gen = DeviceBuffer(n_leaves * T.GLEN)
_ffi.check(L.pfq_synth_genomes_device(d_gen.ptr, n_leaves, T.GLEN, 0x5EED0000, None)); synchronize()
ids = [f"G{i:05d}" for i in range(n_leaves)]
gt = BloomTree.build_balanced_device(d_gen.ptr, T.GLEN, n_leaves, ids, T.K, T.NBITS, T.H, T.SEEDS[0], T.SEEDS[1], 0.001, 5000000)
genomes = d_gen.to_numpy().reshape(n_leaves, T.GLEN)
print("genome bytes", set(np.unique(genomes).tolist()))
ot = T._oracle_copy(gt, ids)
rng = np.random.default_rng(1)
for err in (0.0, 0.01):
    seq, off = T._reads(genomes, rng, 4000, err)
    for thr in (0.3, 1.0):
        wc, wh = T._oracle_hits(ot, seq, off, thr)
        c, h, st = T._gpu_hits(gt, seq, off, thr)
        print("err", err, "thr", thr, "oracle hits", len(wh), "gpu hits", len(h), "equal", np.array_equal(h, wh), c == wc)
