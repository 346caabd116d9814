#!/usr/bin/env python3
"""Makes tests/golden/examples/: a small subset of the reference's example DATA (BASELINE config 1 plumbing case)
plus the expected outputs for it.

Inputs are data files of the reference repository (examples/genomes/viral_genome_dir/*.fna,
examples/test_reads/*.fq — SURVEY.md §2 row 17), subsampled so the fixture stays < 1 MB.  The reference binary
cannot be built in this image (Rust), so the expected CLASSIFICATION.csv / per-read hit lists are produced by the
CPU oracle (oracle/), on the balanced synthetic tree `phage_filter build-balanced` makes with the README's default
parameters (k=20, fpr 0.001, largest genome 1 000 000 -> 14 377 587 bits, 10 hashes; README.md:101,107).
Run from the repo root in the build container:  python tests/golden/make_examples_subset.py
"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/examples"
OUT = os.path.join(ROOT, "tests", "golden", "examples")
SOURCES = ["NC_022067.1", "NC_022086.1", "NC_022329.1", "NC_022331.2", "NC_022335.1"]
SEEDS = (0x0123456789ABCDEF, 0xFEDCBA9876543210)


def fasta_records(path):
    rid, seq = None, []
    for line in open(path):
        if line.startswith(">"):
            if rid is not None:
                yield rid, header, "".join(seq)
            header, rid, seq = line.rstrip("\n"), line[1:].split()[0], []
        else:
            seq.append(line.strip())
    if rid is not None:
        yield rid, header, "".join(seq)


def main():
    from oracle import pfq_oracle as orc
    os.makedirs(os.path.join(OUT, "genomes"), exist_ok=True)
    os.makedirs(os.path.join(OUT, "reads"), exist_ok=True)
    files = sorted(glob.glob(os.path.join(REF, "genomes", "viral_genome_dir", "*.fna")))
    by_id = {}
    for f in files:
        for rid, header, seq in fasta_records(f):
            by_id[rid] = (f, header, seq)
    others = sorted((len(v[2]), k) for k, v in by_id.items() if k not in SOURCES)[:7]
    chosen = SOURCES + [k for _, k in others]
    for rid in chosen:
        f, header, seq = by_id[rid]
        with open(os.path.join(OUT, "genomes", os.path.basename(f)), "w") as o:
            o.write(header + "\n")
            for i in range(0, len(seq), 80):
                o.write(seq[i:i + 80] + "\n")
    for name, step in (("sim_reads_c10000_n5_e0.01.fq", 10), ("sim_reads_c10000_n5_e0.0.fq", 20)):
        lines = open(os.path.join(REF, "test_reads", name)).read().split("\n")
        with open(os.path.join(OUT, "reads", name), "w") as o:
            for r in range(0, (len(lines) // 4), step):
                o.write("\n".join(lines[4 * r:4 * r + 4]) + "\n")

    # expected outputs from the oracle: genomes in the order the CLI's ReadQueue yields them (sorted file names,
    # consumed from the back, file_parser.rs:238), one leaf per record
    gfiles = sorted(glob.glob(os.path.join(OUT, "genomes", "*.fna")))[::-1]
    ids, seqs = [], []
    for f in gfiles:
        for rid, _, seq in fasta_records(f):
            ids.append(rid)
            seqs.append(seq.encode())
    nbits = orc.needed_bits(0.001, 1000000)
    tree = orc.build_balanced_tree(seqs, ids, 20, nbits, orc.optimal_num_hashes(nbits, 1000000), SEEDS[0], SEEDS[1], 0.001, 1000000)
    rfiles = sorted(glob.glob(os.path.join(OUT, "reads", "*.fq")))[::-1]
    rids, reads = [], []
    for f in rfiles:
        lines = open(f).read().split("\n")
        for r in range(len(lines) // 4):
            rids.append(lines[4 * r][1:].split()[0])
            reads.append(lines[4 * r + 1].encode())
    expected = {}
    for thr in (1.0, 0.7, 0.3):
        for v in range(tree.n_nodes):
            tree.mapped_reads[v] = 0
        hits, _, _ = orc.query_batch(tree, reads, thr, threads=8)
        mapped = sorted({rids[r] for r, _ in hits})
        expected[str(thr)] = {"classification_csv": tree.classification_csv(), "n_hits": len(hits), "n_mapped_ids": len(mapped)}
    json.dump({"_source": __doc__, "genome_order": ids, "nbits": nbits, "n_reads": len(reads), "expected": expected},
              open(os.path.join(OUT, "expected.json"), "w"), indent=1)
    print("wrote", OUT, {k: (v["n_hits"], v["n_mapped_ids"]) for k, v in expected.items()})


if __name__ == "__main__":
    main()
