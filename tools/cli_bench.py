"""End-to-end rate of `phage_filter query` (process start, database load, FASTQ ingest, classification, output) on the
GPU box: SURVEY §8f rows 1-2.  Builds a balanced 64-leaf database of the BASELINE config-2 shape on the GPU, writes
synthetic 150 bp FASTQ (50 % positive) and times the CLI for several worker counts, with and without POS/NEG output
and from gzip.  Prints one JSON line per run.

    python tools/cli_bench.py [--reads 16000000] [--threads 1,4,16] [--threshold 1.0] [--block 100000] [--workdir /tmp/pfq_cli_bench]
"""
import argparse
import gzip
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CLI = os.environ.get("PFQ_CLI_BIN") or os.path.join(ROOT, "phagefilter_amd", "phage_filter")   # (PFQ_CLI_BIN: A/B of two builds)


def write_fastq(path, reads_np, first_id):
    """Vectorised FASTQ writer: fixed-width ids so that a block of records is one 2-D byte array."""
    n, L = reads_np.shape
    ids = np.char.zfill(np.arange(first_id, first_id + n).astype("U10"), 10).astype("S10")
    rec = np.empty((n, 1 + 1 + 10 + 6 + 1 + L + 3 + L + 1), dtype=np.uint8)
    c = 0

    def put(b):
        nonlocal c
        rec[:, c:c + len(b)] = np.frombuffer(b, dtype=np.uint8)
        c += len(b)

    put(b"@r")
    rec[:, c:c + 10] = ids.view(np.uint8).reshape(n, 10)
    c += 10
    put(b" len=x\n")
    rec[:, c:c + L] = reads_np
    c += L
    put(b"\n+\n")
    rec[:, c:c + L] = ord("I")
    c += L
    put(b"\n")
    assert c == rec.shape[1]
    with open(path, "ab") as f:
        f.write(rec.tobytes())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=16_000_000)
    ap.add_argument("--threads", default="1,4,16")
    ap.add_argument("--workdir", default="/tmp/pfq_cli_bench")
    ap.add_argument("--leaves", type=int, default=64)
    ap.add_argument("--threshold", default="1.0", help="-f of every query run")
    ap.add_argument("--block", default="100000", help="-b of every query run")
    ap.add_argument("--only", default="", help="'posneg': only the run with both outputs (environment variables reach the CLI)")
    ap.add_argument("--devices", default="", help="comma-separated device lists to run as well, ';'-separated (e.g. '0,0' = two replicas on GPU 0)")
    a = ap.parse_args()
    import torch
    from phagefilter_amd import BloomTree, _ffi

    L = _ffi.lib()
    shutil.rmtree(a.workdir, ignore_errors=True)
    os.makedirs(a.workdir)
    n_g, glen, k, h, nbits = a.leaves, 50000, 21, 10, 71887936
    dg = torch.empty(n_g * glen, dtype=torch.uint8, device="cuda")
    _ffi.check(L.pfq_synth_genomes_device(dg.data_ptr(), n_g, glen, 0x5EED0000, None))
    torch.cuda.synchronize()
    ids = [f"G{i:05d}" for i in range(n_g)]
    t0 = time.time()
    gt = BloomTree.build_balanced_device(dg.data_ptr(), glen, n_g, ids, k, nbits, h, 0x0123456789ABCDEF, 0xFEDCBA9876543210)
    db = os.path.join(a.workdir, "db")
    os.makedirs(db)
    gt.save(db)
    gt.close()
    print(f"# database: {n_g} leaves, built+saved in {time.time() - t0:.1f} s", flush=True)
    fq = os.path.join(a.workdir, "reads.fq")
    t0 = time.time()
    step = 2_000_000
    for first in range(0, a.reads, step):
        n = min(step, a.reads - first)
        dr = torch.empty(n * 150, dtype=torch.uint8, device="cuda")
        _ffi.check(L.pfq_synth_reads_device(dr.data_ptr(), first, n, 150, dg.data_ptr(), glen, n_g, 0x5EED1234, None))
        torch.cuda.synchronize()
        write_fastq(fq, dr.cpu().numpy().reshape(n, 150), first)
    size = os.path.getsize(fq)
    print(f"# {a.reads} reads, {size / 1e9:.2f} GB FASTQ written in {time.time() - t0:.1f} s", flush=True)
    del dg
    torch.cuda.empty_cache()

    def run(label, reads_path, n_reads, threads, extra=()):
        out = os.path.join(a.workdir, "out")
        env = dict(os.environ, PFQ_INGEST_TIMING="1")
        t0 = time.time()
        p = subprocess.run([CLI, "query", "-r", reads_path, "-o", out, "-d", db, "-t", str(threads), "-b", a.block, "-f", a.threshold, *extra],
                           capture_output=True, text=True, env=env)
        wall = time.time() - t0
        assert p.returncode == 0, p.stderr
        loop = [l for l in p.stderr.splitlines() if l.startswith("query loop")]
        ingest = [l for l in p.stderr.splitlines() if l.startswith("ingest:")]
        outl = [l for l in p.stderr.splitlines() if l.startswith("output:")]
        cpul = [l for l in p.stderr.splitlines() if l.startswith("cpu:")]
        csv = open(os.path.join(out, "CLASSIFICATION.csv")).read().splitlines()
        print(json.dumps({"run": label, "threads": threads, "reads": n_reads, "whole_process_s": round(wall, 3),
                          "whole_process_reads_per_s": round(n_reads / wall), "query_loop": loop[0] if loop else None,
                          "ingest": ingest[0] if ingest else None, "output": outl[0] if outl else None, "cpu": cpul[0] if cpul else None, "classified": sum(int(l.split(",")[1]) for l in csv)}), flush=True)

    tmax = max(int(x) for x in a.threads.split(","))
    if a.only == "posneg":
        for rep in range(2):
            run("fastq pos+neg output", fq, a.reads, tmax, ("--pos-filter", "--neg-filter"))
        shutil.rmtree(a.workdir, ignore_errors=True)
        return
    for t in [int(x) for x in a.threads.split(",")]:
        run("fastq counts-only", fq, a.reads, t)
    run("fastq pos+neg output", fq, a.reads, tmax, ("--pos-filter", "--neg-filter"))
    for devs in [d for d in a.devices.split(";") if d]:
        run(f"fastq counts-only --devices {devs}", fq, a.reads, tmax, ("--devices", devs))
        run(f"fastq pos+neg output --devices {devs}", fq, a.reads, tmax, ("--devices", devs, "--pos-filter", "--neg-filter"))
    # gzip: a directory of 8 parts (streams inflate side by side), 1/4 of the reads
    gzdir = os.path.join(a.workdir, "gz")
    os.makedirs(gzdir)
    n_gz = a.reads // 4
    rec_bytes = size // a.reads
    with open(fq, "rb") as f:
        for part in range(8):
            chunk = f.read(rec_bytes * (n_gz // 8))
            with gzip.open(os.path.join(gzdir, f"part{part}.fq.gz"), "wb", compresslevel=1) as g:
                g.write(chunk)
    run("8 x fastq.gz counts-only", gzdir, (n_gz // 8) * 8, tmax)
    run("8 x fastq.gz counts-only", gzdir, (n_gz // 8) * 8, 1)
    shutil.rmtree(a.workdir, ignore_errors=True)


if __name__ == "__main__":
    main()
