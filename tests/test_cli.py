"""The process-level drop-in seam: `phage_filter query` (main.rs:249-376) on the example-data fixture
(tests/golden/examples, BASELINE config 1 plumbing case).  Expected outputs come from the CPU oracle; POS/NEG
files are compared as multisets with genome lists as sets (record order and HashSet order are unspecified in
the reference)."""
import glob
import gzip
import json
import os
import shutil
import subprocess

import pytest

from oracle import pfq_format as fmt
from oracle import pfq_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "tests", "golden", "examples")
CLI = os.path.join(ROOT, "phagefilter_amd", "phage_filter")
SEEDS = (0x0123456789ABCDEF, 0xFEDCBA9876543210)


def fasta_records(path):
    rid, seq = None, []
    for line in open(path):
        if line.startswith(">"):
            if rid is not None:
                yield rid, "".join(seq)
            rid, seq = line[1:].split()[0], []
        else:
            seq.append(line.strip())
    if rid is not None:
        yield rid, "".join(seq)


def fastq_records(path):
    lines = open(path).read().split("\n")
    for r in range(len(lines) // 4):
        yield lines[4 * r][1:].split()[0], lines[4 * r + 1], lines[4 * r + 3]


def example_tree():
    ids, seqs = [], []
    for f in sorted(glob.glob(os.path.join(EX, "genomes", "*.fna")))[::-1]:
        for rid, seq in fasta_records(f):
            ids.append(rid)
            seqs.append(seq.encode())
    nbits = orc.needed_bits(0.001, 1000000)
    return orc.build_balanced_tree(seqs, ids, 20, nbits, orc.optimal_num_hashes(nbits, 1000000), SEEDS[0], SEEDS[1], 0.001, 1000000)


def example_reads():
    out = []
    for f in sorted(glob.glob(os.path.join(EX, "reads", "*.fq")))[::-1]:
        out += list(fastq_records(f))
    return out


def test_oracle_reproduces_examples_golden():
    """CPU: the committed expected outputs are what the oracle computes today (regression pin)."""
    gold = json.load(open(os.path.join(EX, "expected.json")))
    t = example_tree()
    assert [t.tax_id[v] for v in t.leaves_dfs()] == gold["genome_order"]
    reads = example_reads()
    assert len(reads) == gold["n_reads"]
    hits, _, _ = orc.query_batch(t, [r[1].encode() for r in reads], 1.0, threads=4)
    assert t.classification_csv() == gold["expected"]["1.0"]["classification_csv"]
    assert len(hits) == gold["expected"]["1.0"]["n_hits"]


def expected_filtering(t, reads, thr, block):
    """POS/NEG records under the reference's per-block ResultMap semantics (main.rs:334-368, result_map.rs)."""
    for v in range(t.n_nodes):
        t.mapped_reads[v] = 0
    hits, _, _ = orc.query_batch(t, [r[1].encode() for r in reads], thr, threads=4)
    per_read = {}
    for r, v in hits:
        per_read.setdefault(r, set()).add(t.tax_id[v])
    pos, neg = [], []
    for b0 in range(0, len(reads), block):
        rm = {}
        for i in range(b0, min(len(reads), b0 + block)):
            if i in per_read:
                rm.setdefault(reads[i][0], set()).update(per_read[i])
        for i in range(b0, min(len(reads), b0 + block)):
            rid, seq, qual = reads[i]
            if rid in rm:
                pos.append((rid, frozenset(rm[rid]), seq.upper(), qual))
            else:
                neg.append((rid, frozenset(), seq.upper(), qual))
    return sorted(pos, key=repr), sorted(neg, key=repr)


def parse_filter_file(path, fastq):
    recs = []
    lines = open(path).read().split("\n")
    step = 4 if fastq else 2
    for r in range(len(lines) // step):
        head = lines[step * r]
        assert head[0] == ("@" if fastq else ">")
        if " |" in head:
            rid, genomes = head[1:].split(" |")
            gs = frozenset(g for g in genomes.split(",") if g)
        else:
            rid, gs = head[1:], frozenset()
        recs.append((rid, gs, lines[step * r + 1], lines[step * r + 3] if fastq else None))
    return sorted(recs, key=repr)


@pytest.fixture(scope="module")
def cli_db(gpu, tmp_path_factory):
    db = str(tmp_path_factory.mktemp("cli") / "db")
    out = subprocess.run([CLI, "build-balanced", "--genomes", os.path.join(EX, "genomes"), "--db-path", db],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return db


@pytest.mark.gpu
def test_cli_db_equals_oracle_tree(cli_db):
    t = example_tree()
    u = fmt.read_db(cli_db)
    assert (u.kmer_size, u.nbits, u.num_hashes, u.seed1, u.seed2) == (20, t.nbits, 10, SEEDS[0], SEEDS[1])
    assert u.tax_id == t.tax_id and u.left == t.left and u.right == t.right
    for v in range(t.n_nodes):
        assert (u.bits[u.filter_of[v]] == t.bits[v]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("thr,block", [("1.0", 100), ("0.7", 1000), ("0.3", 7)])
def test_cli_query_examples(cli_db, tmp_path, thr, block):
    gold = json.load(open(os.path.join(EX, "expected.json")))
    outdir = str(tmp_path / "out")
    os.makedirs(outdir)
    open(os.path.join(outdir, "stale.txt"), "w").write("must be deleted")  # main.rs:380-391 wipes the directory
    p = subprocess.run([CLI, "query", "--reads", os.path.join(EX, "reads"), "--out", outdir, "--db-path", cli_db,
                        "--filter-threshold", thr, "--cache-size", "1", "--block-size-reads", str(block), "--threads", "4",
                        "--pos-filter", "--neg-filter"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert "Querying reads..." in p.stdout and p.stdout.strip().endswith("Finished.")
    assert sorted(os.listdir(outdir)) == ["CLASSIFICATION.csv", "NEG_FILTERING.fq", "POS_FILTERING.fq"]
    assert open(os.path.join(outdir, "CLASSIFICATION.csv")).read() == gold["expected"][thr]["classification_csv"]
    pos, neg = expected_filtering(example_tree(), example_reads(), float(thr), block)
    assert parse_filter_file(os.path.join(outdir, "POS_FILTERING.fq"), True) == pos
    assert parse_filter_file(os.path.join(outdir, "NEG_FILTERING.fq"), True) == [(a, b, c, d) for a, b, c, d in neg]


@pytest.mark.gpu
def test_cli_formats_gz_fasta_single_file_and_depth(cli_db, tmp_path):
    reads = example_reads()[:300]
    fa = tmp_path / "reads.fasta"
    with open(fa, "w") as f:
        for rid, seq, _ in reads:
            f.write(f">{rid} some description\n{seq[:60]}\n{seq[60:]}\n")       # multi-line FASTA
    gz = tmp_path / "reads.fq.gz"
    with gzip.open(gz, "wt") as f:
        for rid, seq, qual in reads:
            f.write(f"@{rid}\n{seq}\n+\n{qual}\n")
    t = example_tree()
    hits, _, _ = orc.query_batch(t, [r[1].encode() for r in reads], 1.0)
    want = t.classification_csv()
    for src, ext in ((fa, "fa"), (gz, "fq")):
        out = str(tmp_path / f"out_{ext}")
        p = subprocess.run([CLI, "query", "-r", str(src), "-o", out, "-d", cli_db, "--pos-filter"], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        assert open(os.path.join(out, "CLASSIFICATION.csv")).read() == want
        assert os.path.exists(os.path.join(out, f"POS_FILTERING.{ext}")) and not os.path.exists(os.path.join(out, f"NEG_FILTERING.{ext}"))
    # --search-depth prunes the tree: leaves become Internal_Node_* (bloom_tree.rs:302-330)
    t = example_tree()
    t.prune(2)
    orc.query_batch(t, [r[1].encode() for r in reads], 1.0)
    out = str(tmp_path / "out_depth")
    p = subprocess.run([CLI, "query", "-r", str(gz), "-o", out, "-d", cli_db, "--search-depth", "2", "-F", "fastq"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert "If using a search depth, use a filtering flag" in p.stdout and "Search depth settings: 2" in p.stdout
    assert open(os.path.join(out, "CLASSIFICATION.csv")).read() == t.classification_csv()
    assert os.listdir(out) == ["CLASSIFICATION.csv"]


@pytest.mark.gpu
def test_cli_errors_exit_like_a_panic(cli_db, tmp_path):
    p = subprocess.run([CLI, "query", "-r", os.path.join(EX, "reads"), "-o", str(tmp_path / "o"), "-d", str(tmp_path / "nodb")],
                       capture_output=True, text=True)
    assert p.returncode == 101 and "tree.bin" in p.stderr
    p = subprocess.run([CLI, "query", "-r", os.path.join(EX, "reads"), "-d", cli_db], capture_output=True, text=True)
    assert p.returncode == 101 and "--out" in p.stderr


@pytest.mark.gpu
def test_cli_build_and_add_match_oracle(tmp_path):
    """`phage_filter build` / `add` (main.rs:148-247): same database as the oracle's greedy insertion, record by
    record in input order (directory files from the back of the sorted list), then queryable."""
    import numpy as np
    rng = np.random.default_rng(7)

    def dna(n):
        return bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n).astype(np.uint8))

    fam = [dna(900) for _ in range(4)]
    recs = []
    for i in range(14):
        g = bytearray(fam[i % 4])
        for _ in range(i):
            g[int(rng.integers(0, len(g)))] = ord("ACGT"[int(rng.integers(0, 4))])
        recs.append((f"genome{i}", bytes(g)))
    gdir = tmp_path / "genomes"
    gdir.mkdir()
    (gdir / "a.fasta").write_bytes(b"".join(b">%s first batch\n%s\n" % (i.encode(), s) for i, s in recs[:5]))
    (gdir / "b.fna").write_bytes(b"".join(b">%s\n%s\n%s\n" % (i.encode(), s[:400], s[400:]) for i, s in recs[5:9]))
    more = tmp_path / "more.fa"
    more.write_bytes(b"".join(b">%s\n%s\n" % (i.encode(), s) for i, s in recs[9:]))
    order = recs[5:9] + recs[:5]          # b.fna is popped first
    db = tmp_path / "db"
    p = subprocess.run([CLI, "build", "-g", str(gdir), "-d", str(db), "-k", "15", "-f", "0.01", "-l", "2000", "--seed1", "5",
                        "--seed2", "10"], capture_output=True, text=True)
    assert p.returncode == 0 and "Building the SBT..." in p.stdout and "Finished." in p.stdout, p.stderr
    ot = orc.build_greedy_tree([s for _, s in order], [i for i, _ in order], 15, 0.01, 2000, 5, 10)
    assert fmt.encode_tree(fmt.read_db(str(db))) == fmt.encode_tree(ot)
    p = subprocess.run([CLI, "add", "-g", str(more), "-d", str(db)], capture_output=True, text=True)
    assert p.returncode == 0 and "Adding new genomes to the SBT..." in p.stdout, p.stderr
    for i, s in recs[9:]:
        orc.greedy_insert(ot, s, i)
    orc.renumber_preorder(ot)
    lt = fmt.read_db(str(db))
    assert fmt.encode_tree(lt) == fmt.encode_tree(ot)
    for v in range(ot.n_nodes):
        assert np.array_equal(lt.bits[lt.filter_of[v]], ot.bits[ot.filter_of[v]])
    # random seeds when none are given: two builds differ in their seeds, both are valid databases
    p = subprocess.run([CLI, "build", "-g", str(more), "-d", str(tmp_path / "db2"), "-k", "15", "-l", "2000"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    t2 = fmt.read_db(str(tmp_path / "db2"))
    assert (t2.seed1, t2.seed2) != (5, 10) and len(t2.leaves_dfs()) == 5


@pytest.mark.gpu
def test_cli_accepts_benchmark_harness_invocations(tmp_path):
    """the exact command lines of benchmarking/bench/tools/phage_filter.py:79-86 (build) and :105-116
    (query), and the two output files its parse_output reads (:41-66) in the way it reads them — including the
    harness's habit of opening POS_FILTERING.fa, which exists only for FASTA reads."""
    import numpy as np
    from collections import Counter
    rng = np.random.default_rng(11)

    def dna(n):
        return bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n).astype(np.uint8))

    genomes = [(f"NC_{1000 + i}.1", dna(3000)) for i in range(6)]
    gdir = tmp_path / "genomes"
    gdir.mkdir()
    for name, seq in genomes:
        (gdir / f"{name}.fna").write_bytes(b">%s some virus\n%s\n" % (name.encode(), seq))
    # simulated reads are named <genome>_<n> (bench/simulate_reads.py) — parse_output strips the last '_' field
    reads = []
    for i in range(400):
        g = int(rng.integers(0, 8))
        if g < 6:
            o = int(rng.integers(0, 3000 - 100))
            reads.append((f"{genomes[g][0]}_{i}", genomes[g][1][o:o + 100].decode()))
        else:
            reads.append((f"random_{i}", dna(100).decode()))
    fa = tmp_path / "reads.fa"
    fa.write_text("".join(f">{rid}\n{seq}\n" for rid, seq in reads))
    db, out = str(tmp_path / "db"), str(tmp_path / "out")
    k, theta, threads = 20, 1.0, 4
    build_cmd = [CLI, "build", "--genomes", str(gdir), "--db-path", db, "--kmer-size", f"{k}", "--cache-size", f"{100}",
                 "--false-pos-rate", f"{0.00001}", "--largest-genome", f"{500000}", "--threads", f"{threads}"]
    p = subprocess.run(build_cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    t = fmt.read_db(db)
    assert (t.kmer_size, t.nbits, t.num_hashes) == (20, orc.needed_bits(0.00001, 500000), orc.optimal_num_hashes(orc.needed_bits(0.00001, 500000), 500000))
    assert sorted(t.tax_id[v] for v in t.leaves_dfs()) == sorted(n for n, _ in genomes)
    for depth, filter_reads in [(None, False), (None, True), (1, True)]:
        run_cmd = [CLI, "query", "--reads", str(fa), "--db-path", db, "--filter-threshold", f"{theta}", "--cache-size", f"{1}",
                   "--block-size-reads", f"{1000}", "--out", out, "--threads", f"{threads}"]
        if depth is not None:
            run_cmd += ["--search-depth", f"{depth}"]
        if filter_reads:
            run_cmd += ["--pos-filter"]
        p = subprocess.run(run_cmd, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        # expected, from the oracle on the database the CLI built
        ot = fmt.read_db(db)
        if depth is not None:
            ot.prune(depth)
        hits, _, _ = orc.query_batch(ot, [s.encode() for _, s in reads], theta, threads=4)
        if not filter_reads:                                   # parse_output, CLASSIFICATION.csv branch (:52-66)
            name2counts = {}
            for line in open(out + "/CLASSIFICATION.csv"):
                name, count = line.strip("\n").split(",")
                name2counts[name] = int(count)
            assert name2counts == {n: c for n, c in ot.leaf_counts() if c > 0}
            assert sum(name2counts.values()) >= sum(1 for rid, _ in reads if not rid.startswith("random"))
        else:                                                  # parse_output, filter_reads branch (:41-51)
            read_counter = Counter()
            with open(out + "/POS_FILTERING.fa", "r") as f:
                for line in f:
                    if line[0] == ">":
                        read_counter["_".join(line.strip(">").split(" ")[0].split("_")[:-1])] += 1
            want = Counter("_".join(reads[r][0].split("_")[:-1]) for r in sorted({r for r, _ in hits}))
            assert read_counter == want and sum(want.values()) >= 290


# ---------------------------------------------------------------------------------------------------------------
# several replicas behind the CLI (`--devices`): dealer / merger checked against the single-device run byte for byte
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("filtering", [False, True])
def test_cli_devices_two_replicas_equal_single_device(cli_db, tmp_path, filtering):
    """`--devices 0,0`: two replicas of the database on one GPU, each fed by its own host thread; segments (counts
    only) or batches (POS/NEG filtering) are dealt to whichever replica asks next, the per-genome counts are combined
    by pfq_trees_allreduce_counts and replica 0 writes CLASSIFICATION.csv.  Every output file must equal the
    single-device run's."""
    base = [CLI, "query", "--reads", os.path.join(EX, "reads"), "--db-path", cli_db, "--filter-threshold", "0.7",
            "--block-size-reads", "64", "--threads", "4"] + (["--pos-filter", "--neg-filter"] if filtering else [])
    outs = {}
    # small segments / batches so that both replicas get work
    env = dict(os.environ, PFQ_INGEST_CHUNK_BYTES="20000", PFQ_INGEST_TIMING="1", PFQ_CLI_BATCH_READS="128")
    for name, extra in (("one", []), ("two", ["--devices", "0,0"]), ("env", [])):
        out = str(tmp_path / name)
        e = dict(env, PFQ_DEVICES="0,0,0") if name == "env" else env
        p = subprocess.run(base + ["--out", out] + extra, capture_output=True, text=True, env=e)
        assert p.returncode == 0, p.stderr
        assert ("on 1 device(s)" if name == "one" else f"on {2 if name == 'two' else 3} device(s)") in p.stderr
        outs[name] = {f: open(os.path.join(out, f), "rb").read() for f in sorted(os.listdir(out))}
    gold = json.load(open(os.path.join(EX, "expected.json")))
    assert outs["one"]["CLASSIFICATION.csv"].decode() == gold["expected"]["0.7"]["classification_csv"]
    assert outs["two"] == outs["one"] and outs["env"] == outs["one"]
    assert len(outs["one"]) == (3 if filtering else 1)
    p = subprocess.run(base + ["--out", str(tmp_path / "bad"), "--devices", "0,7777"], capture_output=True, text=True)
    assert p.returncode == 101 and "device" in p.stderr


@pytest.mark.gpu
def test_cli_devices_on_a_database_with_stored_counts(cli_db, tmp_path):
    """`query --devices 0,0` on a database whose tree.bin holds non-zero mapped_reads (saved after a query): the stored
    counts are reported once, as on one device (replicas x stored before)."""
    import shutil
    from phagefilter_amd import BloomTree, pack_reads
    db = str(tmp_path / "db_counted")
    shutil.copytree(cli_db, db)
    t = BloomTree.load(db)
    reads = [s.encode() for _, s, _ in example_reads()][:300]
    seq, off = pack_reads(reads)
    t.query_packed(seq, off, 0.7)
    assert sum(c for _, c in t.get_leaf_counts()) > 0
    t.save(db)
    t.close()
    base = [CLI, "query", "--reads", os.path.join(EX, "reads"), "--db-path", db, "--filter-threshold", "0.7", "--threads", "4"]
    env = dict(os.environ, PFQ_INGEST_CHUNK_BYTES="20000")
    outs = {}
    for name, extra in (("one", []), ("two", ["--devices", "0,0"])):
        out = str(tmp_path / name)
        p = subprocess.run(base + ["--out", out] + extra, capture_output=True, text=True, env=env)
        assert p.returncode == 0, p.stderr
        outs[name] = open(os.path.join(out, "CLASSIFICATION.csv")).read()
    gold = json.load(open(os.path.join(EX, "expected.json")))
    assert outs["one"] != gold["expected"]["0.7"]["classification_csv"]    # the stored counts are in
    assert outs["two"] == outs["one"]


@pytest.mark.gpu
def test_cli_block_size_zero_processes_nothing(cli_db, tmp_path):
    """--block-size-reads 0: the reference's first block is empty (file_parser.rs:252-270), its loop never runs
    (main.rs:334-368): outputs are created empty, nothing is parsed — not even a malformed file is noticed."""
    bad = tmp_path / "bad.fq"
    bad.write_text("@r1\nACGT\n+\n")                              # truncated record
    for src in (os.path.join(EX, "reads"), str(bad)):
        out = str(tmp_path / "out")
        p = subprocess.run([CLI, "query", "-r", src, "-o", out, "-d", cli_db, "-b", "0", "--pos-filter", "--neg-filter"],
                           capture_output=True, text=True)
        assert p.returncode == 0 and p.stdout.strip().endswith("Finished."), p.stderr
        assert {f: os.path.getsize(os.path.join(out, f)) for f in os.listdir(out)} == \
            {"CLASSIFICATION.csv": 0, "POS_FILTERING.fq": 0, "NEG_FILTERING.fq": 0}


# ---------------------------------------------------------------------------------------------------------------
# BASELINE config 1 end to end through the CLI's own greedy `build`
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_config1_greedy_build_then_query_examples(tmp_path):
    """`phage_filter build` on the example genomes with the README's defaults (k = 20, fpr 0.001, largest genome 10^6;
    main.rs:148-200: one leaf per FASTA record, greedy placement) and `phage_filter query` on the example reads: the
    database equals the oracle's restatement of BloomTree::insert record by record, CLASSIFICATION.csv and the POS/NEG
    files equal the oracle's query on that tree."""
    db, out = str(tmp_path / "db"), str(tmp_path / "out")
    p = subprocess.run([CLI, "build", "--genomes", os.path.join(EX, "genomes"), "--db-path", db, "--seed1", str(SEEDS[0]),
                        "--seed2", str(SEEDS[1])], capture_output=True, text=True)
    assert p.returncode == 0 and "Finished." in p.stdout, p.stderr
    ids, seqs = [], []
    for f in sorted(glob.glob(os.path.join(EX, "genomes", "*.fna")))[::-1]:       # files are popped from the back
        for rid, seq in fasta_records(f):
            ids.append(rid)
            seqs.append(seq.encode())
    ot = orc.build_greedy_tree(seqs, ids, 20, 0.001, 1000000, SEEDS[0], SEEDS[1])
    lt = fmt.read_db(db)
    assert fmt.encode_tree(lt) == fmt.encode_tree(ot)
    for v in range(ot.n_nodes):
        assert (lt.bits[lt.filter_of[v]] == ot.bits[ot.filter_of[v]]).all(), v
    reads = example_reads()
    for thr, block in (("1.0", 100), ("0.5", 1000)):
        p = subprocess.run([CLI, "query", "--reads", os.path.join(EX, "reads"), "--out", out, "--db-path", db,
                            "--filter-threshold", thr, "--block-size-reads", str(block), "--pos-filter", "--neg-filter"],
                           capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        pos, neg = expected_filtering(ot, reads, float(thr), block)
        assert open(os.path.join(out, "CLASSIFICATION.csv")).read() == ot.classification_csv()
        assert len(ot.classification_csv()) > 0
        assert parse_filter_file(os.path.join(out, "POS_FILTERING.fq"), True) == pos
        assert parse_filter_file(os.path.join(out, "NEG_FILTERING.fq"), True) == neg
