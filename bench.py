#!/usr/bin/env python3
"""bench.py — reads/s classified (150 bp reads vs a 1024-leaf SBT) on N MI355X, one process per GPU.

A "step" is one pass of the hot path (pfq_query_batch_device: k-mer hashing + Bloom frontier + leaf certificates)
over one batch of synthetic 150 bp reads that is already resident in HBM.  Workload = BASELINE.json config 2/3
(SURVEY.md §8d): balanced 1024-leaf SBT over 50 kbp random genomes, k=21, nbits=71 887 936, 10 hashes, fixed
seeds; reads 50 % positive (uniform leaf / offset / strand, error-free) and 50 % uniform random, theta = 1.0.
With N > 1 every rank holds a replica of the tree, classifies its own shard of the reads (weak scaling, no
data-path collective) and the per-genome counts are combined by ONE RCCL all-reduce inside the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) including `roofline` for the dominant kernel
(HIP-event time measured live on the launch stream) and `cpu_baseline` (the CPU oracle in reference-faithful
mode on the host cores, rank 0, N = 1 only; a reported baseline, not the target).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

K, NBITS, NUM_HASHES = 21, 71887936, 10
SEEDS = (0x0123456789ABCDEF, 0xFEDCBA9876543210)
GENOME_SEED, READ_SEED = 0x5EED0000, 0x5EED1234
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--leaves", type=int, default=1024)
    ap.add_argument("--reads-per-step", type=int, default=8 * 1024 * 1024, help="per GPU")
    ap.add_argument("--genome-len", type=int, default=50000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--threshold", type=float, default=1.0)
    ap.add_argument("--path", type=int, default=-1, help="-1 auto, 0 direct kernel, 1 bucketed")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget; 0 disables it")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node N")
    # Rehearsal knobs (not used by the driver): PFQ_BENCH_SAME_GPU=1 puts every rank on device 0 and
    # PFQ_BENCH_BACKEND=gloo replaces RCCL, so the N > 1 code path can be exercised on a one-GPU box.
    dev_index = 0 if os.environ.get("PFQ_BENCH_SAME_GPU") == "1" else local_rank
    backend = os.environ.get("PFQ_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)

    from phagefilter_amd import BloomTree, _ffi
    from phagefilter_amd.dist import all_reduce_counts
    L = _ffi.lib()
    n_g, glen, rl, B = args.leaves, args.genome_len, args.read_len, args.reads_per_step
    ids = [f"G{i:05d}" for i in range(n_g)]

    # ---- database: genomes generated on the device, SBT built on the device (replica per GPU)
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
    tree = BloomTree.build_balanced_device(genomes.data_ptr(), glen, n_g, ids, K, NBITS, NUM_HASHES, SEEDS[0], SEEDS[1],
                                           0.001, 5000000, device=dev_index)
    tree.set_path(args.path)

    # ---- reads: every (step, rank) gets its own slice of the global read index space, resident in HBM
    n_batches = min(args.steps + args.warmup, 16)
    reads = torch.empty(n_batches * B * rl + 64, dtype=torch.uint8, device=dev)
    src_genomes = genomes
    if os.environ.get("PFQ_BENCH_ALL_NEGATIVE") == "1":  # experiment: reads drawn from genomes that are NOT in the tree
        src_genomes = torch.empty(n_g * glen, dtype=torch.uint8, device=dev)
        _ffi.check(L.pfq_synth_genomes_device(src_genomes.data_ptr(), n_g, glen, GENOME_SEED + 0x100000, None))
    for b in range(n_batches):
        first = (b * world + rank) * B
        _ffi.check(L.pfq_synth_reads_device(reads.data_ptr() + b * B * rl, first, B, rl, src_genomes.data_ptr(), glen, n_g,
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
    counts = torch.zeros(n_g, dtype=torch.int64, device=dev)

    def barrier() -> None:
        if world > 1:
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
    all_reduce_counts(counts)  # one RCCL all-reduce over xGMI (8 KiB at 1024 leaves); no-op at N = 1
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    prof = tree.profile_end()
    st = tree.last_stats()
    total_reads = world * args.steps * B
    total_hits = int(counts.sum().item())

    result = None
    if rank == 0:
        kern = {"k_classify": prof.classify_ms, "bucket(scan+scatter)": prof.bucket_ms, "k_tile_plan+k_tile_bin": prof.bin_ms,
                "k_tile_test": prof.test_ms, "k_verify": prof.verify_ms, "k_finalize": prof.finalize_ms}
        calls = max(prof.calls, 1)
        read_bytes = B * rl
        cert_bytes = int(st.algorithmic_bytes) - read_bytes
        # The certificates (the algorithmic bytes beyond the reads themselves) are produced by one stage: in tile mode
        # k_tile_bin + k_tile_test together (probes binned, then tested out of LDS), else k_verify, else (direct path)
        # k_classify itself.  The roofline object describes that stage.
        if st.path == 1 and st.tile_mode:
            dom, avg_ms, alg = "k_tile_bin+k_tile_test (certificate stage)", (prof.bin_ms + prof.test_ms) / calls, cert_bytes
        elif st.path == 1:
            dom, avg_ms, alg = "k_verify", prof.verify_ms / calls, cert_bytes
        else:
            dom, avg_ms, alg = "k_classify", prof.classify_ms / calls, int(st.algorithmic_bytes)
        achieved = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic, per_kernel_traffic = None, {}
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):  # HBM bytes per launch from a separate rocprofv3 --pmc pass of this same workload
            try:
                tj = json.load(open(tpath))
                if (tj.get("reads_per_step") == B and tj.get("leaves") == n_g and args.threshold == 1.0 and rl == 150
                        and not any(os.environ.get(v) for v in ("PFQ_BENCH_FAMILY", "PFQ_BENCH_READ_ERRORS", "PFQ_BENCH_ALL_NEGATIVE"))):
                    per_kernel_traffic = tj.get("hbm_bytes_per_launch", {})
            except Exception:
                per_kernel_traffic = {}
        if per_kernel_traffic:
            if st.path == 1 and st.tile_mode:
                traffic = per_kernel_traffic.get("k_tile_bin", 0) + per_kernel_traffic.get("k_tile_test", 0)
            elif st.path == 1:
                traffic = per_kernel_traffic.get("k_verify_rec")
        # per-kernel view: time, the algorithmic bytes of the work the kernel itself performs, measured HBM traffic
        per_kernel = [
            {"kernel": "k_classify", "ms": prof.classify_ms / calls, "algorithmic_bytes": read_bytes if st.path == 1 else int(st.algorithmic_bytes),
             "hbm_bytes": per_kernel_traffic.get("k_classify")},
            {"kernel": "k_tile_plan+k_tile_bin", "ms": prof.bin_ms / calls, "algorithmic_bytes": 0, "hbm_bytes": per_kernel_traffic.get("k_tile_bin")},
            {"kernel": "k_tile_test", "ms": prof.test_ms / calls, "algorithmic_bytes": cert_bytes if (st.path == 1 and st.tile_mode) else 0,
             "hbm_bytes": per_kernel_traffic.get("k_tile_test")},
            {"kernel": "k_verify", "ms": prof.verify_ms / calls, "algorithmic_bytes": cert_bytes if (st.path == 1 and not st.tile_mode) else 0,
             "hbm_bytes": per_kernel_traffic.get("k_verify_rec")},
        ]
        kernels_ms = sum(kern.values()) / calls
        whole_gbs = int(st.algorithmic_bytes) / (kernels_ms * 1e-3) / 1e9 if kernels_ms > 0 else 0.0
        whole_traffic = sum(v for v in per_kernel_traffic.values() if v) if per_kernel_traffic else None
        result = {
            "metric": "reads/sec classified (150 bp, 1024-leaf SBT) at 1/2/4/8 MI355X; bit-exact vs CPU",
            "value": total_reads / elapsed, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{B} synthetic {rl} bp reads per step per GPU (50% positive / 50% random, theta={args.threshold}) "
                                   f"vs balanced {n_g}-leaf SBT, k={K}, nbits={NBITS}, {NUM_HASHES} hashes",
                       "reads_per_step_per_gpu": B, "read_len": rl, "leaves": n_g, "k": K, "nbits": NBITS,
                       "num_hashes": NUM_HASHES, "threshold": args.threshold,
                       "parallelism": f"reads sharded x{world}, tree replicated per GPU, one RCCL all-reduce of per-genome counts"},
            # SURVEY §8d's contract figure: A(r) summed over the reads of one step (one launch sequence) / the time of the
            # kernels of that step (HIP events on the launch stream).  The path is a sequence of kernels none of which
            # dominates, so the object describes the sequence; `certificate_stage` is the stage that produces the
            # certificates (the algorithmic bytes beyond the reads themselves) on its own.
            "roofline": {"bound": "hbm", "achieved": whole_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": whole_gbs / HBM_PEAK_GBS, "traffic": whole_traffic,
                         "kernel": "whole step: " + " + ".join(k for k, v in kern.items() if v > 0),
                         "avg_launch_ms": kernels_ms, "algorithmic_bytes_per_launch": int(st.algorithmic_bytes),
                         "certificate_stage": {"kernel": dom, "achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
                                               "traffic": traffic, "avg_launch_ms": avg_ms,
                                               "algorithmic_bytes_per_launch": alg}},
            "query_path": "bucketed(screen+L2-sliced verify)" if st.path == 1 else "direct",
            "n_slices": int(st.n_slices), "tile_mode": int(st.tile_mode), "fallback_pairs": int(st.n_fallback_pairs),
            "tile_chunks": int(st.n_chunks), "tile_entries": int(st.tile_entries),
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

    # ---- CPU baseline: oracle in reference-faithful mode on the host cores (rank 0, N = 1 only)
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        result["cpu_baseline"] = cpu_baseline(tree, reads, n_g, ids, B, rl, args, np, torch)
    if rank == 0:
        print(json.dumps(result), flush=True)
    tree.close()
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(tree, reads, n_g, ids, B, rl, args, np, torch):
    """Oracle (`oracle/`, kind "port": a C restatement of the reference CPU path — the reference is Rust and cannot
    be built here) timed on a bounded prefix of step 0's reads, same tree (copied back from HBM), on this box's share of the host cores."""
    from oracle import pfq_oracle as orc
    # the CPU share of this box: 16 cores per visible GPU (a one-GPU box is a slice of a 256-core host), unless told
    cores = int(os.environ.get("PFQ_BENCH_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16 * max(1, torch.cuda.device_count()))))
    ot = orc.balanced_topology(ids, K, NBITS, NUM_HASHES, SEEDS[0], SEEDS[1], 0.001, 5000000, alloc_bits=False)
    ot.bits = np.empty((ot.n_nodes, ot.n_words), dtype=np.uint64)
    for v in range(ot.n_nodes):
        ot.bits[v] = tree.node_filter(v)
    chunk, done, secs, probes = 20000, 0, 0.0, 0
    t_wall = time.perf_counter()
    gpu_check = None
    while done + chunk <= B and (time.perf_counter() - t_wall) < args.cpu_seconds:
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
    return {"value": done / secs if secs > 0 else 0.0, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": f"first {done} reads of step 0 (same tree copied back from HBM); oracle in reference-faithful mode "
                      f"(DFS with per-node re-hash and per-k-mer early exit, query.rs:99-158), query phase only",
            "probes_per_read": probes / max(done, 1), "gpu_parity_on_sample": gpu_check}


if __name__ == "__main__":
    main()
