#!/usr/bin/env python3
"""bench.py — reads/s classified (150 bp reads vs a 1024-leaf SBT) on N MI355X, one process per GPU.

A "step" is one pass of the hot path (pfq_query_batch_device: k-mer hashing + Bloom frontier + leaf certificates)
over one batch of synthetic 150 bp reads that is already resident in HBM.  Workload = BASELINE.json config 3
(SURVEY.md §8d): balanced 1024-leaf SBT over 50 kbp random genomes, k=21, nbits=71 887 936, 10 hashes, fixed
seeds; reads 50 % positive (uniform leaf / offset / strand, error-free) and 50 % uniform random, theta = 1.0.

  python bench.py --gpus N --steps K --warmup W
      N > 1 without a launcher: this process starts the N ranks itself (torch.distributed.run, one process per GPU,
      before anything touches a GPU) and exits with their status.  Under a launcher (WORLD_SIZE set) it is one rank.
      Every rank holds a replica of the tree and classifies its own shard of the reads (weak scaling, no data-path
      collective); the per-genome counts are combined by ONE RCCL all-reduce inside the timed region (config 4).
  python bench.py --gpus N --subtree-depth D --leaves 16384      (BASELINE config 5, 2^D == N)
      the tree is subtree-sharded: rank r builds and holds only shard r of the depth-D frontier, EVERY rank classifies
      ALL reads against its shard, and one all-reduce of the zero-padded count vector gives the whole tree's counts.
      (N = 1 with --subtree-index i: one shard alone, i.e. one rank's work of the sharded job.)

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (contract figure of SURVEY §8d over the
kernel time measured live with HIP events on the launch stream, plus the measured-traffic view `hbm_utilisation`) and
`cpu_baseline` (the CPU oracle in reference-faithful mode on the host cores, rank 0, N = 1 only; a reported baseline,
not the target).  The GPU result is checked before the line is printed — against the oracle on a sample and against the
workload's ground truth on everything timed; a mismatch exits non-zero without a line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

K, NBITS, NUM_HASHES = 21, 71887936, 10
SEEDS = (0x0123456789ABCDEF, 0xFEDCBA9876543210)
GENOME_SEED, READ_SEED = 0x5EED0000, 0x5EED1234
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
EXPERIMENT_VARS = ("PFQ_BENCH_FAMILY", "PFQ_BENCH_READ_ERRORS", "PFQ_BENCH_ALL_NEGATIVE")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--leaves", type=int, default=1024)
    ap.add_argument("--reads-per-step", type=int, default=8 * 1024 * 1024, help="per GPU (subtree mode: per step, seen by every rank)")
    ap.add_argument("--genome-len", type=int, default=50000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--threshold", type=float, default=1.0)
    ap.add_argument("--path", type=int, default=-1, help="-1 auto, 0 direct kernel, 1 bucketed")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget; 0 disables the timing (not the parity check)")
    ap.add_argument("--k", type=int, default=K, help="scenario runs only: k-mer size (the metric's workload is the default)")
    ap.add_argument("--nbits", type=int, default=NBITS, help="scenario runs only: filter bits")
    ap.add_argument("--hashes", type=int, default=NUM_HASHES, help="scenario runs only: hashes per k-mer")
    ap.add_argument("--subtree-depth", type=int, default=0, help="config 5: shards = nodes of this depth, one per rank (2^D == N)")
    ap.add_argument("--subtree-index", type=int, default=-1, help="N = 1 only: run this one shard of the depth-D frontier")
    return ap.parse_args()


def self_launch(args) -> int:
    """N > 1 typed without a launcher: start the ranks as fresh child processes.  This parent never touches a GPU."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def read_plan(np, seed, first, count, n_genomes):
    """(is_positive, source leaf) of reads first..first+count from the generator's definition (oracle/pfq_oracle.c)."""
    def sm(x):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))
    with np.errstate(over="ignore"):
        r = np.arange(first, first + count, dtype=np.uint64)
        w0 = sm(sm(np.uint64(seed)) + np.uint64(8) * r)
    return (w0 & np.uint64(1)).astype(bool), ((w0 >> np.uint64(8)) % np.uint64(n_genomes)).astype(np.int64)


def main() -> None:
    global K, NBITS, NUM_HASHES
    args = parse_args()
    K, NBITS, NUM_HASHES = args.k, args.nbits, args.hashes
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        raise SystemExit(self_launch(args))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # Rehearsal knobs (not used by the driver): PFQ_BENCH_SAME_GPU=1 puts every rank on device 0 and
    # PFQ_BENCH_BACKEND=gloo replaces RCCL, so the N > 1 code path can be exercised on a one-GPU box.
    # (RCCL refuses two ranks on one device — "Duplicate GPU detected" — so the same-GPU rehearsal defaults to gloo.)
    same_gpu = os.environ.get("PFQ_BENCH_SAME_GPU") == "1"
    dev_index = 0 if same_gpu else local_rank
    backend = os.environ.get("PFQ_BENCH_BACKEND", "gloo" if same_gpu else "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # (PFQ_BENCH_FORCE_PG=1: a process group even for one rank, so that every RCCL call of the N > 1 path — init with a
    # device id, barriers, the all-reduces and the all-gather — can be exercised on a one-GPU box)
    use_pg = world > 1 or os.environ.get("PFQ_BENCH_FORCE_PG") == "1"
    if use_pg:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)

    from phagefilter_amd import BloomTree, _ffi
    from phagefilter_amd.dist import all_reduce_counts, pad_and_reduce
    L = _ffi.lib()
    n_g, glen, rl, B = args.leaves, args.genome_len, args.read_len, args.reads_per_step
    ids = [f"G{i:05d}" for i in range(n_g)]
    experiment = [v for v in EXPERIMENT_VARS if os.environ.get(v)]

    # ---- sharding mode
    subtree = args.subtree_depth > 0
    shard_index = 0
    if subtree:
        n_shards = 1 << args.subtree_depth
        if world == 1 and args.subtree_index >= 0:
            shard_index = args.subtree_index
        elif n_shards == world:
            shard_index = rank
        else:
            raise SystemExit(f"--subtree-depth {args.subtree_depth} makes {n_shards} shards: run with --gpus {n_shards}, "
                             f"or one shard alone with --gpus 1 --subtree-index i")

    # ---- database: genomes generated on the device, SBT (or this rank's subtree shard) built on the device
    t_setup = time.perf_counter()
    genomes = torch.empty(n_g * glen, dtype=torch.uint8, device=dev)
    _ffi.check(L.pfq_synth_genomes_device(genomes.data_ptr(), n_g, glen, GENOME_SEED, None))
    torch.cuda.synchronize()
    fam = int(os.environ.get("PFQ_BENCH_FAMILY", "0"))
    if fam > 1:  # experiment: families of `fam` related genomes (PFQ_BENCH_DIVERGENCE substitutions per base, default
        # 0.001) so that a positive read passes several leaves
        div = float(os.environ.get("PFQ_BENCH_DIVERGENCE", "0.001"))
        g2 = genomes.view(n_g, glen)
        base = g2[(torch.arange(n_g, device=dev) // fam) * fam].clone()
        gen = torch.Generator(device=dev)
        gen.manual_seed(12345)
        mut = torch.rand(n_g, glen, device=dev, generator=gen) < div
        alt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (n_g, glen), device=dev, generator=gen)]
        g2.copy_(torch.where(mut, alt, base))
        del base, mut, alt
        torch.cuda.synchronize()
    if subtree:
        tree = BloomTree.build_balanced_subtree_device(genomes.data_ptr(), glen, n_g, ids, K, NBITS, NUM_HASHES, SEEDS[0], SEEDS[1],
                                                       args.subtree_depth, shard_index, 0.001, 5000000, device=dev_index)
    else:
        tree = BloomTree.build_balanced_device(genomes.data_ptr(), glen, n_g, ids, K, NBITS, NUM_HASHES, SEEDS[0], SEEDS[1],
                                               0.001, 5000000, device=dev_index)
    tree.set_path(args.path)
    info = tree.info()
    n_local, first_leaf = int(info.n_leaves), int(info.shard_first_leaf)

    # ---- reads, resident in HBM.  Read sharding: every (step, rank) gets its own slice of the global read index space.
    # Subtree sharding: every rank regenerates the SAME reads (all reads meet every shard).
    n_batches = min(args.steps + args.warmup, 16)
    reads = torch.empty(n_batches * B * rl + 64, dtype=torch.uint8, device=dev)
    src_genomes = genomes
    if os.environ.get("PFQ_BENCH_ALL_NEGATIVE") == "1":  # experiment: reads drawn from genomes that are NOT in the tree
        src_genomes = torch.empty(n_g * glen, dtype=torch.uint8, device=dev)
        _ffi.check(L.pfq_synth_genomes_device(src_genomes.data_ptr(), n_g, glen, GENOME_SEED + 0x100000, None))

    def first_read(b: int) -> int:
        return b * B if subtree else (b * world + rank) * B

    for b in range(n_batches):
        _ffi.check(L.pfq_synth_reads_device(reads.data_ptr() + b * B * rl, first_read(b), B, rl, src_genomes.data_ptr(), glen, n_g,
                                            READ_SEED, None))
    err = float(os.environ.get("PFQ_BENCH_READ_ERRORS", "0"))
    if err > 0:  # experiment: substitution errors in the reads (thresholds below 1 are made for these)
        gen = torch.Generator(device=dev)
        gen.manual_seed(777)
        for b in range(n_batches):
            view = reads[b * B * rl:(b + 1) * B * rl]
            mut = torch.rand(view.numel(), device=dev, generator=gen) < err
            alt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (view.numel(),), device=dev, generator=gen)]
            view.copy_(torch.where(mut, alt, view))
            del mut, alt
    off = torch.arange(B + 1, dtype=torch.int64, device=dev) * rl
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream

    def step(i: int) -> None:
        b = i % n_batches
        tree.query_device(reads.data_ptr() + b * B * rl, off.data_ptr(), B, B * rl, args.threshold, stream)

    step(0)  # builds the HBM layout on first use (not timed)
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t_setup
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    tree.reset_counts()
    counts = torch.zeros(max(n_local, 1), dtype=torch.int64, device=dev)

    def barrier() -> None:
        if use_pg:
            if backend == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()

    # ---- timed region: exactly K steps + the single all-reduce of per-genome counts
    tree.profile_begin(args.steps)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    tree.export_counts(counts.data_ptr(), stream)
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0          # this rank's own classification time
    local_counts = counts[:n_local].clone()
    t_red = time.perf_counter()
    if subtree:
        total_counts = pad_and_reduce(counts[:n_local], first_leaf, n_g)   # one all-reduce of the zero-padded vector
    else:
        total_counts = all_reduce_counts(counts)   # one RCCL all-reduce over xGMI (8 KiB at 1024 leaves); no-op at N = 1
    torch.cuda.synchronize()
    allreduce_ms = (time.perf_counter() - t_red) * 1e3
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank_s = [t_local]
    if use_pg:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        mine = torch.tensor([t_local], dtype=torch.float64, device=dev)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank_s = [float(g.item()) for g in gathered]

    prof = tree.profile_end()
    st = tree.last_stats()
    # every read of the job is counted once: with read sharding the ranks' reads add up, with subtree sharding all ranks
    # classify the same reads
    total_reads = args.steps * B if subtree else world * args.steps * B
    total_hits = int(total_counts.sum().item())

    # ---- checks on everything that was timed (all ranks): the reduction is the sum of the ranks' counters, every positive
    # read hit its source leaf, and what else was hit stays within the Bloom false-positive rate
    local_sum = torch.tensor([int(local_counts.sum().item())], dtype=torch.int64, device=dev)
    if use_pg:
        dist.all_reduce(local_sum, op=dist.ReduceOp.SUM)
    problems = []
    if int(local_sum.item()) != total_hits:
        problems.append(f"reduced counts sum to {total_hits}, the ranks' own counters to {int(local_sum.item())}")
    if not experiment and args.threshold == 1.0 and rl >= K:
        expect = np.zeros(n_g, dtype=np.int64)
        for i in range(args.steps):
            for r in (range(1) if subtree else range(world)):
                b = (args.warmup + i) % n_batches
                f = b * B if subtree else (b * world + r) * B
                pos, leaf = read_plan(np, READ_SEED, f, B, n_g)
                expect += np.bincount(leaf[pos], minlength=n_g)
        got = total_counts.cpu().numpy()[:n_g]
        if subtree and world == 1:   # one shard alone: only its own leaves are held
            expect, got = expect[first_leaf:first_leaf + n_local], got[first_leaf:first_leaf + n_local]
        if not (got >= expect).all():
            problems.append("a positive read did not hit its source leaf")
        elif int(got.sum() - expect.sum()) > max(64, total_reads // 50000):
            problems.append(f"{int(got.sum() - expect.sum())} hits beyond the positives' source leaves (Bloom false positives should be rare)")

    result = None
    if rank == 0:
        kern = {"k_classify": prof.classify_ms, "bucket(scan+scatter)": prof.bucket_ms, "k_tile_plan+k_tile_bin": prof.bin_ms,
                "k_tile_test": prof.test_ms, "k_verify": prof.verify_ms, "k_finalize": prof.finalize_ms}
        calls = max(prof.calls, 1)
        read_bytes = B * rl
        cert_bytes = int(st.algorithmic_bytes) - read_bytes
        # HBM bytes per launch from a separate rocprofv3 --pmc pass of this same workload (tools/pmc_traffic.py), quoted only
        # when it was measured on the library sources that are running now
        per_kernel_traffic, traffic_stale = {}, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                same_workload = (tj.get("reads_per_step") == B and tj.get("leaves") == n_g and tj.get("threshold", 1.0) == args.threshold
                                 and rl == 150 and not experiment and not subtree)
                if same_workload:
                    traffic_stale = tj.get("source_stamp") != _ffi.source_stamp()
                    if not traffic_stale:
                        per_kernel_traffic = tj.get("hbm_bytes_per_launch", {})
            except Exception:
                per_kernel_traffic = {}
        ms = {"k_classify": prof.classify_ms / calls, "k_tile_plan+k_tile_bin": prof.bin_ms / calls, "k_tile_test": prof.test_ms / calls,
              "k_verify": prof.verify_ms / calls, "bucket(scan+scatter)": prof.bucket_ms / calls, "k_finalize": prof.finalize_ms / calls}
        hbm = {"k_classify": (per_kernel_traffic.get("k_classify", 0) + per_kernel_traffic.get("k_tail_records", 0)) or None,
               "k_tile_plan+k_tile_bin": (per_kernel_traffic.get("k_tile_bin", 0) + per_kernel_traffic.get("k_tile_plan", 0)) or None,
               "k_tile_test": per_kernel_traffic.get("k_tile_test"), "k_verify": per_kernel_traffic.get("k_verify_rec"),
               "bucket(scan+scatter)": per_kernel_traffic.get("k_bucket_scatter"), "k_finalize": per_kernel_traffic.get("k_finalize")}
        alg = {"k_classify": read_bytes if st.path == 1 else int(st.algorithmic_bytes), "k_tile_plan+k_tile_bin": 0,
               "k_tile_test": cert_bytes if (st.path == 1 and st.tile_mode) else 0,
               "k_verify": cert_bytes if (st.path == 1 and not st.tile_mode) else 0, "bucket(scan+scatter)": 0, "k_finalize": 0}
        per_kernel = [{"kernel": k, "ms": ms[k], "algorithmic_bytes": alg[k], "hbm_bytes": hbm[k],
                       "hbm_utilisation": (hbm[k] / (ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS) if (hbm[k] and ms[k] > 0) else None}
                      for k in ms if ms[k] > 0 or alg[k]]
        kernels_ms = sum(kern.values()) / calls
        whole_gbs = int(st.algorithmic_bytes) / (kernels_ms * 1e-3) / 1e9 if kernels_ms > 0 else 0.0
        whole_traffic = sum(v for v in per_kernel_traffic.values() if v) if per_kernel_traffic else None
        # what ANY exact algorithm for this step must move through HBM at least once: the reads, and — because half the reads
        # hit and their certificates touch every part of their leaf's filter — each leaf filter of this rank once
        must_touch = read_bytes + n_local * ((NBITS + 63) // 64) * 8
        frac_measured = (whole_traffic / (kernels_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (whole_traffic and kernels_ms > 0) else None
        if subtree:
            par = (f"tree subtree-sharded at depth {args.subtree_depth} (shard {shard_index} of {1 << args.subtree_depth} per rank), every rank "
                   f"classifies all reads, one RCCL all-reduce of the zero-padded per-genome counts")
        else:
            par = f"reads sharded x{world}, tree replicated per GPU, one RCCL all-reduce of per-genome counts"
        result = {
            "metric": "reads/sec classified (150 bp, 1024-leaf SBT) at 1/2/4/8 MI355X; bit-exact vs CPU",
            "value": total_reads / elapsed, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if subtree else "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{B} synthetic {rl} bp reads per step per GPU (50% positive / 50% random, theta={args.threshold}) "
                                   f"vs balanced {n_g}-leaf SBT, k={K}, nbits={NBITS}, {NUM_HASHES} hashes"
                                   + (f"; subtree-sharded, {n_local} leaves on this rank" if subtree else ""),
                       "reads_per_step_per_gpu": B, "read_len": rl, "leaves": n_g, "k": K, "nbits": NBITS,
                       "num_hashes": NUM_HASHES, "threshold": args.threshold, "parallelism": par},
            # SURVEY §8d's contract figure: A(r) = L(r) + |hits(r)| * need(r) * num_hashes * 32 B summed over the reads of one
            # step (one launch sequence) / the time of the kernels of that step (HIP events on the launch stream).  The path
            # is a sequence of kernels none of which dominates, so the object describes the sequence.  `traffic` = measured
            # HBM bytes of the same kernels (separate --pmc pass; null when that pass is not of this code), and
            # `hbm_utilisation` = traffic / time / peak: what the step really draws from HBM.  Certificates are tested out
            # of LDS, so the measured traffic is far BELOW the contract's 32 B per probe.
            # `frac` stays the contract figure (it can exceed what the bytes really moved would give: probes are tested out of
            # LDS tiles, 4 B of entry per probe instead of a 32 B sector); `frac_measured` = measured HBM bytes / time / peak is
            # the utilisation to read as "how busy is the HBM", and `must_touch_bytes` the floor of any exact algorithm —
            # a step faster than the contract ceiling still moves at least that.
            "roofline": {"bound": "hbm", "achieved": whole_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": whole_gbs / HBM_PEAK_GBS, "traffic": whole_traffic, "traffic_stale": traffic_stale,
                         "traffic_source": ("stored rocprofv3 --pmc pass of this same code and workload (profiles/pmc_traffic.json, "
                                            "TCC_EA0_RDREQ x 128 B + TCC_EA0_WRREQ x 64 B), not measured in this run") if whole_traffic else None,
                         "frac_measured": frac_measured, "hbm_utilisation": frac_measured,
                         "must_touch_bytes": must_touch,
                         "must_touch_frac": (must_touch / (kernels_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if kernels_ms > 0 else None,
                         "kernel": "whole step: " + " + ".join(k for k, v in kern.items() if v > 0),
                         "avg_launch_ms": kernels_ms, "algorithmic_bytes_per_launch": int(st.algorithmic_bytes),
                         "units_per_launch": B, "algorithmic_bytes_per_unit": int(st.algorithmic_bytes) / B},
            "rccl_ranks": dist.get_world_size() if (use_pg and backend == "nccl") else 0,
            "collective_backend": backend if use_pg else None, "allreduce_ms": allreduce_ms,
            "per_rank_reads_per_s": [args.steps * B / s for s in per_rank_s],
            "query_path": "bucketed(screen + leaf-sorted certificates)" if st.path == 1 else "direct",
            "n_slices": int(st.n_slices), "tile_mode": int(st.tile_mode), "fallback_pairs": int(st.n_fallback_pairs),
            "tile_chunks": int(st.n_chunks), "tile_entries": int(st.tile_entries),
            "leaf_groups": int(st.leaf_groups), "coarse_cols": int(st.coarse_cols), "coarse_probes": int(st.coarse_probes),
            "group_reads_last_step": int(st.group_reads),
            "kernel_ms_per_step": {k: v / calls for k, v in kern.items()}, "per_kernel": per_kernel,
            "hits_total": total_hits, "hits_last_step": int(st.n_hits), "candidates_last_step": int(st.n_candidates),
            "setup_seconds": setup_s,
        }

    # ---- PCIe-inclusive rate: the same block handed over as HOST buffers (pfq_query_batch); never `value`
    if rank == 0 and world == 1:
        h_seq = np.concatenate([reads[:B * rl].cpu().numpy(), np.zeros(16, dtype=np.uint8)])
        h_off = np.arange(B + 1, dtype=np.uint64) * rl
        best = None
        for _ in range(2):
            t_h = time.perf_counter()
            tree.query_packed(h_seq, h_off, args.threshold)
            torch.cuda.synchronize()  # the call returns once the kernels are queued
            dt = time.perf_counter() - t_h
            best = dt if best is None else min(best, dt)
        result["host_buffers_reads_per_s"] = B / best
        # ---- the same HBM-resident block with PFQ_WANT_HITS (what POS/NEG filtering needs, main.rs:345-361): synchronous, the
        # per-read hit lists come back to the host as CSR; never `value`
        best = None
        for _ in range(2):
            t_h = time.perf_counter()
            offs, leaves = tree.query_device_hits(reads.data_ptr(), off.data_ptr(), B, B * rl, args.threshold, stream)
            dt = time.perf_counter() - t_h
            best = dt if best is None else min(best, dt)
        result["want_hits_reads_per_s"] = B / best
        result["want_hits_pairs"] = int(offs[-1])

    # ---- parity vs the oracle on a sample (rank 0, always) + the CPU baseline timing (N = 1, --cpu-seconds > 0)
    if rank == 0:
        check = oracle_check_and_baseline(tree, reads, n_g, ids, B, rl, args, np, torch, subtree, shard_index,
                                          time_it=(world == 1 and args.cpu_seconds > 0))
        if check.get("gpu_parity_on_sample") not in ("ok", "skipped"):
            problems.append(f"GPU counts differ from the oracle's on the sample: {check.get('gpu_parity_on_sample')}")
        result["parity"] = {"oracle_sample": check.get("gpu_parity_on_sample"), "sample": check.get("parity_sample"),
                            "timed_region_ground_truth": "ok" if not problems else "FAILED"}
        if world == 1 and args.cpu_seconds > 0 and "value" in check:
            result["cpu_baseline"] = {k: v for k, v in check.items() if k != "parity_sample"}
    bad = torch.tensor([len(problems)], dtype=torch.int64, device=dev)
    if use_pg:
        dist.all_reduce(bad, op=dist.ReduceOp.SUM)
    if problems:
        print(f"[rank {rank}] bench.py: WRONG RESULTS, no bench line: " + "; ".join(problems), file=sys.stderr, flush=True)
    if rank == 0 and int(bad.item()) == 0:
        print(json.dumps(result), flush=True)
    tree.close()
    if use_pg:
        dist.destroy_process_group()
    if int(bad.item()):
        raise SystemExit(1)


def oracle_check_and_baseline(tree, reads, n_g, ids, B, rl, args, np, torch, subtree, shard_index, time_it):
    """Oracle (`oracle/`, kind "port": a C restatement of the reference CPU path — the reference is Rust and cannot be
    built here) on the same tree, copied back from HBM.  Always: the GPU's counts of the first 20 000 reads of step 0 are
    compared with the oracle's (skipped only when the tree's filters would not fit a host copy).  With `time_it`: the
    oracle in reference-faithful mode timed on a bounded prefix of step 0's reads on this box's share of the host cores."""
    from oracle import pfq_oracle as orc
    # the CPU share of this box: 16 cores per visible GPU (a one-GPU box is a slice of a 256-core host), unless told
    cores = int(os.environ.get("PFQ_BENCH_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16 * max(1, torch.cuda.device_count()))))
    ot = orc.balanced_topology(ids, K, NBITS, NUM_HASHES, SEEDS[0], SEEDS[1], 0.001, 5000000, alloc_bits=False)
    first = 0
    if subtree:
        ot, first = orc.subtree_shard(ot, args.subtree_depth, shard_index)
    keep, stack = [], [ot.root]
    while stack:  # the nodes this tree holds (a shard: its subtree + the chain of ancestors)
        v = stack.pop()
        keep.append(v)
        stack += [c for c in (ot.left[v], ot.right[v]) if c >= 0]
    if len(keep) * ot.n_words * 8 > 24 << 30:
        return {"gpu_parity_on_sample": "skipped", "parity_sample": f"{len(keep)} filters do not fit a host copy"}
    ot.bits = np.empty((len(keep), ot.n_words), dtype=np.uint64)
    for row, v in enumerate(keep):
        ot.bits[row] = tree.node_filter(v)
        ot.filter_of[v] = row
    # (PFQ_BENCH_PARITY_READS: scenario runs whose oracle is slow — ten thousand leaves at threshold 0.3 — check fewer reads)
    chunk, done, secs, probes = int(os.environ.get("PFQ_BENCH_PARITY_READS", "20000")), 0, 0.0, 0
    n_parity = chunk
    t_wall = time.perf_counter()
    gpu_check = None
    while done + chunk <= B and (gpu_check is None or (time_it and (time.perf_counter() - t_wall) < args.cpu_seconds)):
        seq = reads[done * rl:(done + chunk) * rl].cpu().numpy()
        seq = np.concatenate([seq, np.zeros(16, dtype=np.uint8)])
        off = np.arange(chunk + 1, dtype=np.uint64) * rl
        before = list(ot.mapped_reads)
        _, p, s = orc.query_batch_packed(ot, seq, off, args.threshold, faithful=True, threads=cores, want_hits=False)
        if gpu_check is None:  # parity of the first chunk: GPU counts vs oracle counts on the same reads
            tree.reset_counts()
            tree.query_packed(seq, off, args.threshold)
            want = [(ot.tax_id[v], ot.mapped_reads[v] - before[v]) for v in ot.leaves_dfs()]
            gpu_check = "ok" if tree.get_leaf_counts() == want else "MISMATCH"
        done += chunk
        secs += s
        probes += p
        chunk = min(chunk * 2, 200000)
    out = {"gpu_parity_on_sample": gpu_check or "skipped", "parity_sample": f"first {n_parity} reads of step 0, per-leaf counts"}
    if time_it:
        out.update({"value": done / secs if secs > 0 else 0.0, "unit": "reads/s", "cores": cores, "kind": "port",
                    "sample": f"first {done} reads of step 0 (same tree copied back from HBM); oracle in reference-faithful mode "
                              f"(DFS with per-node re-hash and per-k-mer early exit, query.rs:99-158), query phase only",
                    "probes_per_read": probes / max(done, 1)})
    return out


if __name__ == "__main__":
    main()
