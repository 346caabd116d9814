// pfq_cli.cpp — `phage_filter query` on top of libpfq's C ABI: the process-level drop-in seam.
//
// Mirrors the query arm of the reference CLI (src/main.rs:100-135 flags, :249-376 flow, :380-404 helpers), the
// FASTA/FASTQ(.gz) ingest of src/file_parser.rs:33-101,:191-344 (bio 2.2.0 readers) and ResultMap
// (src/result_map.rs:9-46).  All classification work happens on the GPU through include/pfq.h; this file only
// parses text, keeps the reference's block bookkeeping and writes CLASSIFICATION.csv / POS_FILTERING.* /
// NEG_FILTERING.*.  `build` / `add` (tree construction) are out of scope; `build-balanced` makes the synthetic
// balanced tree of SURVEY §8d from a genome directory so the query path can be exercised end to end.
#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <map>
#include <mutex>
#include <set>
#include <thread>
#include <string>
#include <vector>

#include "../../include/pfq.h"

namespace {

[[noreturn]] void die(const std::string &msg) {  // the reference panics: message on stderr, exit code 101
    fprintf(stderr, "phage_filter: %s\n", msg.c_str());
    exit(101);
}
void check(int rc) {
    if (rc != PFQ_OK) die(std::string("libpfq: ") + pfq_last_error());
}

// ---------------------------------------------------------------------------------------------------------------
// input files (file_parser.rs:303-344)
// ---------------------------------------------------------------------------------------------------------------
const char *SEQ_EXT[] = {"fa", "fasta", "fna", "fsa", "fas", "fq", "fastq"};
std::string ext_of(const std::string &p) {
    size_t slash = p.find_last_of('/');
    std::string base = slash == std::string::npos ? p : p.substr(slash + 1);
    size_t dot = base.find_last_of('.');
    if (dot == std::string::npos || dot == 0) return "";
    return base.substr(dot + 1);
}
std::string stem_of(const std::string &p) {
    size_t dot = p.find_last_of('.');
    return dot == std::string::npos ? p : p.substr(0, dot);
}
bool is_seq_ext(const std::string &e) {
    for (auto s : SEQ_EXT)
        if (e == s) return true;
    return false;
}
bool has_supported_extension(const std::string &path) {
    std::string e = ext_of(path);
    if (e.empty()) return false;
    if (is_seq_ext(e)) return true;
    if (e == "gz" || e == "gzip") return is_seq_ext(ext_of(stem_of(path)));
    return false;
}
// get_file_names: a file is taken as is; a directory contributes its entries with a supported extension.
// read_dir order is unspecified in the reference; here: sorted by name (and consumed from the back, like pop()).
std::vector<std::string> get_file_names(const std::string &path) {
    struct stat st;
    if (stat(path.c_str(), &st) != 0) die("cannot stat '" + path + "': " + strerror(errno));
    if (S_ISREG(st.st_mode)) return {path};
    std::vector<std::string> out;
    DIR *d = opendir(path.c_str());
    if (!d) die("cannot read directory '" + path + "'");
    while (dirent *e = readdir(d)) {
        std::string name = e->d_name;
        if (name == "." || name == "..") continue;
        std::string full = path + (path.back() == '/' ? "" : "/") + name;
        if (has_supported_extension(full)) out.push_back(full);
    }
    closedir(d);
    std::sort(out.begin(), out.end());
    return out;
}

enum class Fmt { Fasta, Fastq };
enum class FmtOverride { Auto, Fasta, Fastq };

Fmt format_from_extension(const std::string &path) {  // file_parser.rs:69-86
    std::string e = ext_of(path), low = e;
    for (auto &c : low) c = (char)tolower(c);
    std::string eff = (low == "gz" || low == "gzip") ? ext_of(stem_of(path)) : e;
    return (eff == "fq" || eff == "fastq") ? Fmt::Fastq : Fmt::Fasta;
}
Fmt detect_format(const std::string &path, FmtOverride ov) {  // file_parser.rs:33-66
    if (ov == FmtOverride::Fasta) return Fmt::Fasta;
    if (ov == FmtOverride::Fastq) return Fmt::Fastq;
    gzFile f = gzopen(path.c_str(), "rb");  // transparent for plain files, inflates gzip
    if (f) {
        int c = gzgetc(f);
        gzclose(f);
        if (c == '>') return Fmt::Fasta;
        if (c == '@') return Fmt::Fastq;
    }
    return format_from_extension(path);
}

// Line reader over a (possibly gzip-compressed) file.  Lines are handed out as views into the read buffer
// (no per-line allocation); a line that straddles a buffer refill is assembled in `carry`.
struct LineReader {
    gzFile f = nullptr;
    std::vector<char> buf;
    std::string carry;
    size_t pos = 0, len = 0;
    bool eof = false;
    explicit LineReader(const std::string &path) : buf(4 << 20) {
        f = gzopen(path.c_str(), "rb");
        if (!f) die("Failed to open '" + path + "': " + strerror(errno));
        gzbuffer(f, 1 << 20);
    }
    ~LineReader() {
        if (f) gzclose(f);
    }
    // One line without its '\n' as [p, p+n); false at end of file with nothing read.  The view stays valid until
    // the next call.
    bool next(const char *&p, size_t &n) {
        bool use_carry = false;
        while (true) {
            if (pos == len) {
                if (eof) {
                    if (use_carry) { p = carry.data(); n = carry.size(); return true; }
                    return false;
                }
                int got = gzread(f, buf.data(), (unsigned)buf.size());
                if (got < 0) die("read error (corrupt gzip?)");
                if (got == 0) { eof = true; continue; }
                pos = 0;
                len = (size_t)got;
            }
            const char *b = buf.data() + pos;
            const char *nl = (const char *)memchr(b, '\n', len - pos);
            if (nl) {
                if (use_carry) {
                    carry.append(b, nl - b);
                    p = carry.data();
                    n = carry.size();
                } else {
                    p = b;
                    n = (size_t)(nl - b);
                }
                pos += (size_t)(nl - b) + 1;
                return true;
            }
            if (!use_carry) { carry.clear(); use_carry = true; }
            carry.append(b, len - pos);
            pos = len;
        }
    }
};

inline size_t trimmed_len(const char *p, size_t n) {  // str::trim_end
    while (n && isspace((unsigned char)p[n - 1])) --n;
    return n;
}
std::string id_of(const char *h, size_t n) {  // bio Record::id(): first whitespace-delimited token after the marker
    n = trimmed_len(h, n);
    size_t i = 1;
    while (i < n && !isspace((unsigned char)h[i])) ++i;
    return std::string(h + 1, i > 1 ? i - 1 : 0);
}

// One block of reads in the layout the C ABI takes (concatenated bases + n+1 offsets); ids and qualities are kept
// only when POS/NEG filtering needs them (the reference drops them otherwise too, file_parser.rs:202-204,217-220).
struct Batch {
    std::vector<uint8_t> seq;
    std::vector<uint64_t> off{0};
    std::vector<std::string> ids;
    std::vector<char> qual;           // concatenated qualities (same offsets as seq) when has_qual
    std::vector<uint8_t> has_qual;    // per read
    size_t n() const { return off.size() - 1; }
    void clear() {
        seq.clear();
        off.assign(1, 0);
        ids.clear();
        qual.clear();
        has_qual.clear();
    }
};

// bio::io::fasta / fastq record iteration (multi-line sequences; FASTQ qualities read until they cover the
// sequence), appending straight into a Batch.
struct RecordReader {
    LineReader lr;
    Fmt fmt;
    std::string pending;  // a header line already consumed while finishing the previous FASTA record
    bool have_pending = false;
    RecordReader(const std::string &path, Fmt f) : lr(path), fmt(f) {}
    bool next(Batch &b, bool keep) {
        const char *p;
        size_t n;
        std::string header;
        if (have_pending) {
            header.swap(pending);
            have_pending = false;
        } else {
            if (!lr.next(p, n)) return false;
            if (n == 0 && lr.eof && lr.pos == lr.len) return false;
            header.assign(p, n);
        }
        const size_t seq0 = b.seq.size();
        if (fmt == Fmt::Fasta) {
            if (header.empty() || header[0] != '>') die("FASTA: Expected > at record start.");
            while (lr.next(p, n)) {
                if (n && p[0] == '>') {
                    pending.assign(p, n);
                    have_pending = true;
                    break;
                }
                n = trimmed_len(p, n);
                b.seq.insert(b.seq.end(), p, p + n);
            }
            b.off.push_back(b.seq.size());
            if (keep) {
                b.ids.push_back(id_of(header.data(), header.size()));
                b.has_qual.push_back(0);
                b.qual.resize(b.seq.size(), 0);
            }
            return true;
        }
        if (header.empty() || header[0] != '@') die("FASTQ: Expected @ at record start.");
        bool plus = false;
        while (lr.next(p, n)) {
            if (n && p[0] == '+') {
                plus = true;
                break;
            }
            n = trimmed_len(p, n);
            b.seq.insert(b.seq.end(), p, p + n);
        }
        if (!plus) die("FASTQ: Incomplete record (missing '+' line).");
        const size_t slen = b.seq.size() - seq0;
        size_t qlen = 0;
        if (keep) b.qual.resize(seq0, 0);
        while (qlen < slen) {
            if (!lr.next(p, n)) die("FASTQ: Incomplete record (quality shorter than sequence).");
            n = trimmed_len(p, n);
            if (keep) b.qual.insert(b.qual.end(), p, p + n);
            qlen += n;
        }
        if (qlen != slen) die("FASTQ: Unequal length of sequence and quality.");
        b.off.push_back(b.seq.size());
        if (keep) {
            b.ids.push_back(id_of(header.data(), header.size()));
            b.has_qual.push_back(1);
        }
        return true;
    }
};

// ReadQueue (file_parser.rs:227-301): files consumed from the back of the list, records streamed across files.
struct ReadQueue {
    std::vector<std::string> files;
    FmtOverride ov;
    RecordReader *cur = nullptr;
    ReadQueue(const std::string &path, FmtOverride o) : files(get_file_names(path)), ov(o) {}
    ~ReadQueue() { delete cur; }
    Fmt peek_format() const { return files.empty() ? Fmt::Fasta : detect_format(files.back(), ov); }
    bool next(Batch &b, bool keep) {
        while (true) {
            if (!cur) {
                if (files.empty()) return false;
                std::string p = files.back();
                files.pop_back();
                cur = new RecordReader(p, detect_format(p, ov));
            }
            if (cur->next(b, keep)) return true;
            delete cur;
            cur = nullptr;
        }
    }
    // Appends up to max_reads reads (and at most ~max_bytes bases); false when the input is exhausted.
    bool fill(Batch &b, uint64_t max_reads, uint64_t max_bytes, bool keep) {
        while (b.n() < max_reads && b.seq.size() < max_bytes)
            if (!next(b, keep)) return false;
        return true;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// argument parsing (clap surface of main.rs:44-136)
// ---------------------------------------------------------------------------------------------------------------
struct Args {
    std::map<std::string, std::string> val;
    std::set<std::string> flags;
    int verbose = 0, quiet = 0;
};
struct Opt {
    const char *lng;
    char shrt;
    bool takes_value;
};
Args parse(int argc, char **argv, int start, const std::vector<Opt> &opts) {
    Args a;
    auto find_long = [&](const std::string &n) -> const Opt * {
        for (auto &o : opts)
            if (n == o.lng) return &o;
        return nullptr;
    };
    auto find_short = [&](char c) -> const Opt * {
        for (auto &o : opts)
            if (o.shrt && c == o.shrt) return &o;
        return nullptr;
    };
    for (int i = start; i < argc; ++i) {
        std::string s = argv[i];
        if (s.rfind("--", 0) == 0) {
            std::string name = s.substr(2), v;
            bool has_v = false;
            size_t eq = name.find('=');
            if (eq != std::string::npos) {
                v = name.substr(eq + 1);
                name = name.substr(0, eq);
                has_v = true;
            }
            if (name == "verbose") { ++a.verbose; continue; }
            if (name == "quiet") { ++a.quiet; continue; }
            const Opt *o = find_long(name);
            if (!o) die("error: unexpected argument '--" + name + "' found");
            if (!o->takes_value) { a.flags.insert(o->lng); continue; }
            if (!has_v) {
                if (i + 1 >= argc) die("error: a value is required for '--" + name + "'");
                v = argv[++i];
            }
            a.val[o->lng] = v;
        } else if (s.size() >= 2 && s[0] == '-') {
            for (size_t j = 1; j < s.size(); ++j) {
                char c = s[j];
                if (c == 'v') { ++a.verbose; continue; }
                if (c == 'q') { ++a.quiet; continue; }
                const Opt *o = find_short(c);
                if (!o) die(std::string("error: unexpected argument '-") + c + "' found");
                if (!o->takes_value) { a.flags.insert(o->lng); continue; }
                std::string v = s.substr(j + 1);
                if (!v.empty() && v[0] == '=') v = v.substr(1);
                if (v.empty()) {
                    if (i + 1 >= argc) die(std::string("error: a value is required for '-") + c + "'");
                    v = argv[++i];
                }
                a.val[o->lng] = v;
                break;
            }
        } else die("error: unexpected argument '" + s + "' found");
    }
    return a;
}
std::string req(const Args &a, const char *name) {
    auto it = a.val.find(name);
    if (it == a.val.end()) die(std::string("error: the following required arguments were not provided: --") + name);
    return it->second;
}
std::string opt(const Args &a, const char *name, const std::string &def) {
    auto it = a.val.find(name);
    return it == a.val.end() ? def : it->second;
}
uint64_t to_u64(const std::string &s, const char *what) {
    char *e = nullptr;
    errno = 0;
    unsigned long long v = strtoull(s.c_str(), &e, 10);
    if (errno || !e || *e || s.empty() || s[0] == '-') die(std::string("error: invalid value '") + s + "' for '--" + what + "'");
    return v;
}
float to_f32(const std::string &s, const char *what) {
    char *e = nullptr;
    float v = strtof(s.c_str(), &e);
    if (!e || *e || s.empty()) die(std::string("error: invalid value '") + s + "' for '--" + what + "'");
    return v;
}
FmtOverride to_fmt(const std::string &s) {
    if (s == "auto") return FmtOverride::Auto;
    if (s == "fasta") return FmtOverride::Fasta;
    if (s == "fastq") return FmtOverride::Fastq;
    die("error: invalid value '" + s + "' for '--format' [possible values: auto, fasta, fastq]");
}

void rm_rf(const std::string &p) {
    struct stat st;
    if (lstat(p.c_str(), &st) != 0) return;
    if (S_ISDIR(st.st_mode)) {
        DIR *d = opendir(p.c_str());
        if (d) {
            while (dirent *e = readdir(d)) {
                std::string n = e->d_name;
                if (n != "." && n != "..") rm_rf(p + "/" + n);
            }
            closedir(d);
        }
        rmdir(p.c_str());
    } else unlink(p.c_str());
}

int device_from_env() {
    const char *e = getenv("PFQ_DEVICE");
    return e ? atoi(e) : 0;
}

// ---------------------------------------------------------------------------------------------------------------
// query (main.rs:249-376)
// ---------------------------------------------------------------------------------------------------------------
int cmd_query(int argc, char **argv) {
    std::vector<Opt> opts = {{"reads", 'r', true}, {"out", 'o', true}, {"db-path", 'd', true}, {"threads", 't', true},
                             {"block-size-reads", 'b', true}, {"filter-threshold", 'f', true}, {"cache-size", 'c', true},
                             {"search-depth", 0, true}, {"pos-filter", 0, false}, {"neg-filter", 0, false}, {"format", 'F', true}};
    Args a = parse(argc, argv, 2, opts);
    const std::string reads = req(a, "reads"), out = req(a, "out"), db = req(a, "db-path");
    (void)to_u64(opt(a, "threads", "4"), "threads");        // rayon pool size: no meaning on the device path
    (void)to_u64(opt(a, "cache-size", "10"), "cache-size");  // LRU of .bf files: the whole tree is resident in HBM
    uint64_t block = to_u64(opt(a, "block-size-reads", "100"), "block-size-reads");
    const float threshold = to_f32(opt(a, "filter-threshold", "1.0"), "filter-threshold");
    const bool pos = a.flags.count("pos-filter") != 0, neg = a.flags.count("neg-filter") != 0;
    const bool filtering = pos || neg;
    const FmtOverride ov = to_fmt(opt(a, "format", "auto"));

    pfq_tree *tree = nullptr;
    check(pfq_tree_open(db.c_str(), device_from_env(), &tree));
    printf("Querying reads...\n");
    printf("Filtering settings: positive=%s; negative=%s\n", pos ? "true" : "false", neg ? "true" : "false");
    if (a.val.count("search-depth")) {
        uint64_t depth = to_u64(a.val.at("search-depth"), "search-depth");
        if (!filtering) printf("If using a search depth, use a filtering flag (--pos-filter or --neg-filter, or both!)\n");
        printf("Search depth settings: %llu\n", (unsigned long long)depth);
        check(pfq_tree_prune(tree, depth));
    }
    ReadQueue rq(reads, ov);

    // create_and_overwrite_directory (main.rs:380-391): an existing output directory is deleted
    struct stat st;
    if (stat(out.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) rm_rf(out);
    mkdir(out.c_str(), 0777);
    const char *ext = rq.peek_format() == Fmt::Fastq ? "fq" : "fa";
    FILE *pos_f = nullptr, *neg_f = nullptr;
    if (pos && !(pos_f = fopen((out + "/POS_FILTERING." + ext).c_str(), "wb"))) die("cannot create POS_FILTERING in " + out);
    if (neg && !(neg_f = fopen((out + "/NEG_FILTERING." + ext).c_str(), "wb"))) die("cannot create NEG_FILTERING in " + out);

    const char *const *tax = nullptr;
    const uint64_t *cnt = nullptr;
    uint64_t n_leaves = 0;
    check(pfq_leaf_counts(tree, &tax, &cnt, &n_leaves));
    std::vector<std::string> leaf_names(tax, tax + n_leaves);

    // The device processes big batches; ResultMap semantics (ids merged per reference block, cleared per block,
    // main.rs:334-368) are applied per `block` consecutive reads so the outputs do not depend on the batch size.
    if (block == 0) block = 1;  // the reference would loop forever on empty blocks; treat 0 as 1
    const uint64_t batch_reads = std::max<uint64_t>(block, 4u << 20) / block * block;
    // Parsing runs on its own thread, one block ahead of the GPU (double buffering).
    Batch batches[2];
    std::mutex mu;
    std::condition_variable cv;
    int ready[2] = {0, 0};   // 0 = free for the parser, 1 = filled, 2 = filled and last
    std::thread parser([&] {
        bool more = true;
        for (int i = 0; more; i ^= 1) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return ready[i] == 0; });
            }
            batches[i].clear();
            more = rq.fill(batches[i], batch_reads, 3ull << 30, filtering);
            {
                std::lock_guard<std::mutex> lk(mu);
                ready[i] = more ? 1 : 2;
            }
            cv.notify_all();
        }
    });
    for (int i = 0;; i ^= 1) {
        int state;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return ready[i] != 0; });
            state = ready[i];
        }
        Batch &b = batches[i];
        const uint64_t n = b.n();
        if (n) {
            b.seq.resize(b.seq.size() + 16);
            pfq_hits hits{};
            check(pfq_query_batch(tree, b.seq.data(), b.off.data(), n, threshold, filtering ? PFQ_WANT_HITS : 0,
                                  filtering ? &hits : nullptr));
            if (filtering) {
                std::string line;
                for (uint64_t b0 = 0; b0 < n; b0 += block) {
                    const uint64_t b1 = std::min(n, b0 + block);
                    std::map<std::string, std::set<uint32_t>> result_map;  // read id -> leaf set (result_map.rs:20-22)
                    for (uint64_t r = b0; r < b1; ++r)
                        for (uint64_t j = hits.offsets[r]; j < hits.offsets[r + 1]; ++j) result_map[b.ids[r]].insert(hits.leaves[j]);
                    for (uint64_t r = b0; r < b1; ++r) {
                        auto it = result_map.empty() ? result_map.end() : result_map.find(b.ids[r]);
                        FILE *f = it != result_map.end() ? pos_f : neg_f;  // read_mapped
                        if (!f) continue;
                        line.clear();
                        line += b.has_qual[r] ? '@' : '>';  // write_record (main.rs:394-404)
                        line += b.ids[r];
                        if (it != result_map.end()) {       // get_ext_id: "{id} |{g1,g2}" (set order unspecified in the reference)
                            line += " |";
                            bool first = true;
                            for (uint32_t leaf : it->second) {
                                if (!first) line += ',';
                                line += leaf_names[leaf];
                                first = false;
                            }
                        }
                        line += '\n';
                        const size_t s0 = line.size();
                        line.append((const char *)b.seq.data() + b.off[r], b.off[r + 1] - b.off[r]);
                        for (size_t c = s0; c < line.size(); ++c) line[c] = (char)toupper((unsigned char)line[c]);  // main.rs:347-349
                        line += '\n';
                        if (b.has_qual[r]) {
                            line += "+\n";
                            line.append(b.qual.data() + b.off[r], b.off[r + 1] - b.off[r]);
                            line += '\n';
                        }
                        fwrite(line.data(), 1, line.size(), f);
                    }
                }
            }
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            ready[i] = 0;
        }
        cv.notify_all();
        if (state == 2) break;
    }
    parser.join();
    if (pos_f) fclose(pos_f);
    if (neg_f) fclose(neg_f);
    check(pfq_save_leaf_counts(tree, (out + "/CLASSIFICATION.csv").c_str()));
    pfq_tree_close(tree);
    printf("Finished.\n");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// build-balanced: synthetic balanced SBT over a genome directory (NOT the reference's greedy `build`)
// ---------------------------------------------------------------------------------------------------------------
uint64_t needed_bits(float rate, uint32_t items) {  // bloom_filter.rs:354-357, f32 arithmetic
    const float ln2 = 0.693147180559945309417232121458176568f;
    float ln22 = ln2 * ln2;
    float v = roundf((float)items * (logf(1.0f / rate) / ln22));
    return v <= 0 ? 0 : (uint64_t)v;
}
uint32_t optimal_num_hashes(uint64_t bits, uint32_t items) {  // bloom_filter.rs:342-350
    const float ln2 = 0.693147180559945309417232121458176568f;
    float v = roundf((float)bits / (float)items * ln2);
    uint32_t h = v <= 0 ? 0 : (uint32_t)v;
    return std::min<uint32_t>(std::max<uint32_t>(h, 2), 200);
}
int cmd_build_balanced(int argc, char **argv) {
    std::vector<Opt> opts = {{"genomes", 'g', true}, {"db-path", 'd', true}, {"threads", 't', true}, {"kmer-size", 'k', true},
                             {"cache-size", 'c', true}, {"false-pos-rate", 'f', true}, {"largest-genome", 'l', true},
                             {"format", 'F', true}, {"seed1", 0, true}, {"seed2", 0, true}};
    Args a = parse(argc, argv, 2, opts);
    const std::string genomes = req(a, "genomes"), db = req(a, "db-path");
    const uint64_t k = to_u64(opt(a, "kmer-size", "20"), "kmer-size");
    const float fpr = to_f32(opt(a, "false-pos-rate", "0.001"), "false-pos-rate");
    const uint32_t largest = (uint32_t)to_u64(opt(a, "largest-genome", "1000000"), "largest-genome");
    const uint64_t s1 = strtoull(opt(a, "seed1", "81985529216486895").c_str(), nullptr, 0);
    const uint64_t s2 = strtoull(opt(a, "seed2", "18364758544493064720").c_str(), nullptr, 0);
    ReadQueue rq(genomes, to_fmt(opt(a, "format", "auto")));
    Batch g;
    while (rq.next(g, true)) {}  // block size 1 in the reference: one leaf per record (main.rs:148-200)
    std::vector<uint8_t> &seq = g.seq;
    std::vector<uint64_t> &off = g.off;
    std::vector<std::string> &ids = g.ids;
    std::vector<const char *> idp;
    for (auto &s : ids) idp.push_back(s.c_str());
    const uint64_t nbits = needed_bits(fpr, largest);
    pfq_tree *tree = nullptr;
    seq.resize(seq.size() + 16);
    check(pfq_tree_build_balanced(seq.data(), off.data(), ids.size(), idp.data(), k, nbits, optimal_num_hashes(nbits, largest), s1,
                                  s2, fpr, largest, device_from_env(), &tree));
    mkdir(db.c_str(), 0777);
    check(pfq_tree_save(tree, db.c_str()));
    pfq_tree_close(tree);
    printf("Finished.\n");
    return 0;
}

void usage() {
    fprintf(stderr,
            "A fast, simple and memory efficient metagenomic filtering tool. (MI355X query path)\n\n"
            "Usage: phage_filter [-v...|-q...] <COMMAND>\n\nCommands:\n"
            "  query           Queries a set of reads. (ran after building the bloom tree)\n"
            "  build-balanced  Builds a balanced synthetic BloomTree on the GPU (not the reference's greedy build)\n"
            "  build, add      Not part of this build: tree construction stays with the reference binary\n");
}

}  // namespace

int main(int argc, char **argv) {
    // global -v/-q may precede the subcommand (clap-verbosity-flag, main.rs:49-50)
    int first = 1;
    while (first < argc && argv[first][0] == '-' && strcmp(argv[first], "--help") != 0 && strcmp(argv[first], "-h") != 0) ++first;
    if (first >= argc) {
        usage();
        return 2;
    }
    std::string cmd = argv[first];
    // shift so the subcommand sits at argv[1]
    std::vector<char *> av{argv[0], argv[first]};
    for (int i = 1; i < argc; ++i)
        if (i != first) av.push_back(argv[i]);
    if (cmd == "query") return cmd_query((int)av.size(), av.data());
    if (cmd == "build-balanced") return cmd_build_balanced((int)av.size(), av.data());
    if (cmd == "build" || cmd == "add")
        die("`" + cmd + "` (tree construction) is out of scope of the MI355X query path; use the reference binary, or build-balanced");
    usage();
    return 2;
}
