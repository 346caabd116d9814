"""The reference's own benchmark configurations through the CLI exactly as its harness runs it
(/root/reference/benchmarking/bench/tools/phage_filter.py:79-86 build, :105-116 run; benchmarking/config.yaml:1-4:
k = 20, theta = 0.3; simulated reads of 100 bp): whole-process wall time of `phage_filter build` and of
`phage_filter query ... --cache-size 1 --block-size-reads 1000 --pos-filter`, next to the rows BASELINE.md §1 lists
(reference CPU, hardware unstated: 630 genomes / 100 k reads 47.4 s; 99 genomes / 100 k reads 8.20 s; 1 M reads 1 thread
236 - 245 s, 4 threads 117 - 139 s).  Genomes are synthetic (50 kbp, uniform ACGT: the reference's phage genomes are not
in this image), reads 50 % from the genomes with 1 % substitutions, 50 % random.  One JSON line per run.

    python tools/harness_bench.py [--workdir /tmp/pfq_harness]
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "phagefilter_amd", "phage_filter")
COMP = np.arange(256, dtype=np.uint8)
for a, b in zip(b"ACGT", b"TGCA"):
    COMP[a] = b


def write_genomes(d, n, glen, rng):
    os.makedirs(d)
    g = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n, glen))]
    for i in range(n):
        with open(os.path.join(d, f"genome_{i:05d}.fna"), "wb") as f:
            f.write(f">NC_{i:06d}.1 synthetic phage {i}\n".encode())
            for o in range(0, glen, 70):
                f.write(g[i, o:o + 70].tobytes() + b"\n")
    return g


def write_reads(path, genomes, n, rl, rng, err=0.01):
    n_pos = n // 2
    gi = rng.integers(0, genomes.shape[0], n_pos)
    o = rng.integers(0, genomes.shape[1] - rl + 1, n_pos)
    pos = genomes[gi[:, None], o[:, None] + np.arange(rl)[None, :]]
    rc = rng.random(n_pos) < 0.5
    pos[rc] = COMP[pos[rc]][:, ::-1]
    sub = rng.random(pos.shape) < err
    alt = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(sub.sum()))]
    pos[sub] = np.where(alt == pos[sub], COMP[alt], alt)
    reads = np.empty((n, rl), dtype=np.uint8)
    reads[0::2] = pos
    reads[1::2] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n - n_pos, rl))]
    with open(path, "wb") as f:
        for i in range(n):
            f.write(b"@read_%d/1\n" % i + reads[i].tobytes() + b"\n+\n" + b"I" * rl + b"\n")


def timed(cmd):
    t0 = time.monotonic_ns()                                       # (the harness's clock, bench/utils.py:113-116)
    p = subprocess.run(cmd, capture_output=True, text=True)
    dt = (time.monotonic_ns() - t0) * 1e-9
    if p.returncode != 0:
        raise SystemExit(f"{' '.join(cmd)}\n{p.stderr}")
    return dt, p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workdir", default="/tmp/pfq_harness")
    ap.add_argument("--configs", default="99:100000,630:100000,99:1000000", help="genomes:reads,...")
    ap.add_argument("--threads", default="1,4")
    a = ap.parse_args()
    rng = np.random.default_rng(2026)
    shutil.rmtree(a.workdir, ignore_errors=True)
    os.makedirs(a.workdir)
    published = {(630, 100000): "47.4 s (res_filter_memory.csv:37)", (99, 100000): "8.20 s (res_performance_benchmarking.csv:37)",
                 (99, 1000000): "1 thread 235.8 - 244.6 s, 4 threads 116.9 - 139.4 s (res_threading.csv:2-13; tree size of that run unstated)"}
    built = {}
    for cfg in a.configs.split(","):
        n_g, n_r = (int(x) for x in cfg.split(":"))
        if n_g not in built:
            gd, db = os.path.join(a.workdir, f"genomes_{n_g}"), os.path.join(a.workdir, f"db_{n_g}")
            genomes = write_genomes(gd, n_g, 50000, rng)
            # phage_filter.py:79-86
            dt, _ = timed([CLI, "build", "--genomes", gd, "--db-path", db, "--kmer-size", "20", "--threads", "4",
                           "--false-pos-rate", "0.00001", "--largest-genome", "500000"])
            built[n_g] = (db, genomes)
            print(json.dumps({"run": "build", "genomes": n_g, "whole_process_s": round(dt, 3), "genomes_per_s": round(n_g / dt, 1)}), flush=True)
        db, genomes = built[n_g]
        fq = os.path.join(a.workdir, f"reads_{n_g}_{n_r}.fq")
        write_reads(fq, genomes, n_r, 100, rng)
        for t in a.threads.split(","):
            out = os.path.join(a.workdir, "out")
            # phage_filter.py:105-116
            dt, p = timed([CLI, "query", "--reads", fq, "--out", out, "--db-path", db, "--cache-size", "1", "--threads", t,
                           "--filter-threshold", "0.3", "--block-size-reads", "1000", "--pos-filter"])
            n_pos = sum(1 for l in open(os.path.join(out, "POS_FILTERING.fq")) if l.startswith("@read_"))
            n_cls = sum(int(l.split(",")[1]) for l in open(os.path.join(out, "CLASSIFICATION.csv")))
            print(json.dumps({"run": "query --pos-filter", "genomes": n_g, "reads": n_r, "threads": int(t), "whole_process_s": round(dt, 3),
                              "reads_per_s": round(n_r / dt), "pos_reads": n_pos, "classified": n_cls,
                              "reference_published": published.get((n_g, n_r))}), flush=True)
        os.remove(fq)
    shutil.rmtree(a.workdir, ignore_errors=True)


if __name__ == "__main__":
    main()
