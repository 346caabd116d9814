// pfq_cli.cpp — `phage_filter query` on top of libpfq's C ABI: the process-level drop-in seam.
//
// Mirrors the query arm of the reference CLI (src/main.rs:100-135 flags, :249-376 flow, :380-404 helpers), the
// FASTA/FASTQ(.gz) ingest of src/file_parser.rs:33-101,:191-344 (bio 2.2.0 readers) and ResultMap
// (src/result_map.rs:9-46).  All classification work happens on the GPU through include/pfq.h; this file only
// parses text, keeps the reference's block bookkeeping and writes CLASSIFICATION.csv / POS_FILTERING.* /
// NEG_FILTERING.*.  `build` / `add` (main.rs:148-247) drive pfq_tree_create / pfq_tree_insert, the reference's greedy
// placement on the device; `build-balanced` makes the synthetic balanced tree of SURVEY §8d from a genome directory.
#include <dirent.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <random>
#include <set>
#include <thread>
#include <string>
#include <unordered_map>
#include <string_view>
#include <vector>

#include <emmintrin.h>  // SSE2 (x86-64 baseline)
#include <functional>

#include "../../include/pfq.h"

namespace {

// dst[i] = ASCII upper case of src[i] (what `to_uppercase` does to a nucleotide string, main.rs:347-349), 16 bytes at a time
inline void copy_upper(char *dst, const uint8_t *src, size_t n) {
    const __m128i lo = _mm_set1_epi8('a' - 1), hi = _mm_set1_epi8('z' + 1), bit = _mm_set1_epi8(0x20);
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i));
        const __m128i m = _mm_and_si128(_mm_cmpgt_epi8(v, lo), _mm_cmplt_epi8(v, hi));  // (bytes >= 0x80 are negative: no letters)
        _mm_storeu_si128(reinterpret_cast<__m128i *>(dst + i), _mm_sub_epi8(v, _mm_and_si128(m, bit)));
    }
    for (; i < n; ++i) {
        const uint8_t ch = src[i];
        dst[i] = (char)((ch >= 'a' && ch <= 'z') ? ch - 32 : ch);
    }
}

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

// ---------------------------------------------------------------------------------------------------------------
// line sources: a (possibly gzip-compressed) stream, or a range of a memory-mapped plain file
// ---------------------------------------------------------------------------------------------------------------
// Lines are handed out as views (no per-line allocation); with GzLines a line that straddles a buffer refill is
// assembled in `carry`.  A view stays valid until the next call.
struct GzLines {
    gzFile f = nullptr;
    std::vector<char> buf;
    std::string carry;
    size_t pos = 0, len = 0;
    bool eof = false;
    explicit GzLines(const std::string &path) : buf(4 << 20) {
        f = gzopen(path.c_str(), "rb");
        if (!f) die("Failed to open '" + path + "': " + strerror(errno));
        gzbuffer(f, 1 << 20);
    }
    GzLines(const GzLines &) = delete;
    ~GzLines() {
        if (f) gzclose(f);
    }
    // One line without its '\n' as [p, p+n); false at end of file with nothing read.
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
    bool drained() const { return eof && pos == len; }
    uint64_t tell() const { return 0; }  // positions are only used with MemLines
};
struct MemLines {
    const char *base, *cur, *end;
    MemLines(const char *b, uint64_t from, uint64_t to) : base(b), cur(b + from), end(b + to) {}
    bool next(const char *&p, size_t &n) {
        if (cur == end) return false;
        const char *nl = (const char *)memchr(cur, '\n', (size_t)(end - cur));
        p = cur;
        if (nl) {
            n = (size_t)(nl - cur);
            cur = nl + 1;
        } else {
            n = (size_t)(end - cur);
            cur = end;
        }
        return true;
    }
    bool drained() const { return false; }
    uint64_t tell() const { return (uint64_t)(cur - base); }
};

// Bases and offsets of a batch go to the GPU as they are; when a tree is open their buffers are page-locked
// (pfq_host_alloc) so that the copy runs at PCIe rate.  Small blocks and the CPU-only subcommands use malloc.
bool g_pinned = false;  // set once, before the first batch is allocated
template <class T>
struct HostAlloc {
    using value_type = T;
    HostAlloc() = default;
    template <class U>
    HostAlloc(const HostAlloc<U> &) {}
    static bool pinned(size_t n) { return g_pinned && n * sizeof(T) >= (1u << 20); }
    // resize() without a fill: new elements are default-initialised (bytes and offsets are overwritten right away)
    template <class U>
    void construct(U *p) { ::new ((void *)p) U; }
    template <class U, class A0, class... A>
    void construct(U *p, A0 &&a0, A &&...a) { ::new ((void *)p) U(std::forward<A0>(a0), std::forward<A>(a)...); }
    T *allocate(size_t n) {
        void *p = nullptr;
        if (pinned(n)) {
            if (pfq_host_alloc(n * sizeof(T), &p) != PFQ_OK) die(std::string("libpfq: ") + pfq_last_error());
        } else if (!(p = malloc(n * sizeof(T)))) throw std::bad_alloc();
        return (T *)p;
    }
    void deallocate(T *p, size_t n) {
        if (pinned(n)) pfq_host_free(p);
        else free(p);
    }
    template <class U>
    bool operator==(const HostAlloc<U> &) const { return true; }
    template <class U>
    bool operator!=(const HostAlloc<U> &) const { return false; }
};

inline size_t trimmed_len(const char *p, size_t n) {  // str::trim_end
    while (n && isspace((unsigned char)p[n - 1])) --n;
    return n;
}

// One block of reads in the layout the C ABI takes (concatenated bases + n+1 offsets); ids and qualities are kept
// only when POS/NEG filtering needs them (the reference drops them otherwise too, file_parser.rs:202-204,217-220).
struct Segment;
struct Batch {
    std::vector<uint8_t, HostAlloc<uint8_t>> seq;
    std::vector<uint64_t, HostAlloc<uint64_t>> off{0};
    std::vector<char> id_bytes;       // concatenated ids (bio Record::id()) when kept
    std::vector<uint64_t> id_off{0};
    std::vector<char> qual;           // concatenated qualities when kept (bio does not require |qual| == |seq|)
    std::vector<uint64_t> qual_off{0};
    std::vector<uint8_t> has_qual;    // per read
    // A batch assembled for POS/NEG filtering copies only what the device needs (the bases): ids and qualities stay in the
    // parsed segments, which the batch holds until it has been written (append_ref).
    bool external = false;
    std::vector<std::string_view> ext_id, ext_qual;
    std::vector<Segment *> held;
    size_t n() const { return off.size() - 1; }
    void clear() {
        seq.clear();
        off.assign(1, 0);
        id_bytes.clear();
        id_off.assign(1, 0);
        qual.clear();
        qual_off.assign(1, 0);
        has_qual.clear();
        ext_id.clear();
        ext_qual.clear();
        external = false;
    }
    std::string_view quality(size_t r) const {
        return external ? ext_qual[r] : std::string_view(qual.data() + qual_off[r], qual_off[r + 1] - qual_off[r]);
    }
    std::string_view id(size_t r) const {
        return external ? ext_id[r] : std::string_view(id_bytes.data() + id_off[r], id_off[r + 1] - id_off[r]);
    }
    // Record::id(): header[1..].trim_end() up to the first separator — any whitespace in bio's FASTA reader
    // (`splitn(2, char::is_whitespace)`), a blank only in its FASTQ reader (`splitn(2, ' ')`)
    void push_id(const char *h, size_t n, bool fastq) {
        n = trimmed_len(h, n);
        size_t i = 1;
        while (i < n && !(fastq ? h[i] == ' ' : isspace((unsigned char)h[i]))) ++i;
        if (i > 1) id_bytes.insert(id_bytes.end(), h + 1, h + i);
        id_off.push_back(id_bytes.size());
    }
    // reads [r0, r1) of `o` appended to this batch
    void append(const Batch &o, size_t r0, size_t r1, bool keep) {
        auto copy_seq = [&] {
            const uint64_t s0 = o.off[r0], s1 = o.off[r1], base = seq.size();
            seq.insert(seq.end(), o.seq.begin() + s0, o.seq.begin() + s1);
            for (size_t r = r0 + 1; r <= r1; ++r) off.push_back(base + (o.off[r] - s0));
        };
        if (!keep) {
            copy_seq();
            return;
        }
        auto copy_qual = [&] {
            const uint64_t q0 = o.qual_off[r0], q1 = o.qual_off[r1], qb = qual.size();
            qual.insert(qual.end(), o.qual.begin() + q0, o.qual.begin() + q1);
            for (size_t r = r0 + 1; r <= r1; ++r) qual_off.push_back(qb + (o.qual_off[r] - q0));
        };
        // sequences, qualities and ids are separate arrays: large pieces are copied side by side (the assembler thread was
        // the slowest stage of the filtering pipeline: 10 GB of records through one core)
        const bool big = r1 - r0 >= 4096;
        std::thread ts, tq;
        if (big) {
            ts = std::thread(copy_seq);
            tq = std::thread(copy_qual);
        } else {
            copy_seq();
            copy_qual();
        }
        const uint64_t i0 = o.id_off[r0], i1 = o.id_off[r1], ib = id_bytes.size();
        id_bytes.insert(id_bytes.end(), o.id_bytes.begin() + i0, o.id_bytes.begin() + i1);
        for (size_t r = r0 + 1; r <= r1; ++r) id_off.push_back(ib + (o.id_off[r] - i0));
        has_qual.insert(has_qual.end(), o.has_qual.begin() + r0, o.has_qual.begin() + r1);
        if (big) {
            ts.join();
            tq.join();
        }
    }
    // reads [r0, r1) of `o`: bases copied, ids and qualities referenced (the caller keeps `o` alive, see `held`).  Large
    // pieces are handled by four threads side by side: the arrays are sized first, every thread fills its range of reads.
    void append_ref(const Batch &o, size_t r0, size_t r1) {
        external = true;
        const size_t n0 = n(), cnt = r1 - r0;
        const uint64_t s0 = o.off[r0], s1 = o.off[r1], base = seq.size();
        seq.resize(base + (s1 - s0));
        off.resize(n0 + 1 + cnt);
        ext_id.resize(n0 + cnt);
        ext_qual.resize(n0 + cnt);
        has_qual.resize(n0 + cnt);
        auto part = [&](size_t a, size_t b) {  // reads [a, b) of the piece
            memcpy(seq.data() + base + (o.off[r0 + a] - s0), o.seq.data() + o.off[r0 + a], o.off[r0 + b] - o.off[r0 + a]);
            for (size_t i = a; i < b; ++i) {
                off[n0 + 1 + i] = base + (o.off[r0 + i + 1] - s0);
                ext_id[n0 + i] = o.id(r0 + i);
                ext_qual[n0 + i] = o.quality(r0 + i);
                has_qual[n0 + i] = o.has_qual[r0 + i];
            }
        };
        const size_t T = cnt >= 16384 ? 4 : 1;
        std::vector<std::thread> ts;
        for (size_t t = 1; t < T; ++t) ts.emplace_back(part, cnt * t / T, cnt * (t + 1) / T);
        part(0, cnt / T);
        for (auto &t : ts) t.join();
    }
};

// bio 2.2.0 `io::fasta::Reader::read` / `io::fastq::Reader::read` as the reference consumes them through
// `.records()` + `unwrap()` (file_parser.rs:191-224; no `Record::check()`), appending straight into a Batch:
//   FASTA: header line must start with '>'; every following line up to the next '>' line is sequence, trimmed at
//          the end;
//   FASTQ: header line must start with '@'; lines up to the first '+' line are sequence (trimmed, counted); then
//          the SAME NUMBER of lines is read as quality (trimmed); an empty quality is `IncompleteRecord`.  Lengths
//          of sequence and quality are not compared (that is `check()`, which the reference never calls).
// bio is a crates.io dependency that is not vendored in the reference: multi-line and malformed-record behaviour is
// restated from its published source, parity unpinned; the reference's own parser tests (file_parser.rs:410-604)
// only hold ordinary four-line records, which every reading of the rules agrees on.  Malformed input is reported through `err` (the caller decides when
// it becomes fatal: a speculative parse from a guessed record boundary must not kill the process).
template <class Src>
struct RecordParser {
    Src &lr;
    Fmt fmt;
    std::string header, pending;  // pending: a header line already consumed while finishing the previous FASTA record
    bool have_pending = false;
    uint64_t pending_pos = 0;     // where that line starts (MemLines)
    std::string err;
    RecordParser(Src &s, Fmt f) : lr(s), fmt(f) {}
    // Position at which the next record starts (MemLines only).
    uint64_t next_record_pos() const { return have_pending ? pending_pos : lr.tell(); }
    // 1: a record was appended; 0: clean end of input; -1: malformed (message in err)
    int next(Batch &b, bool keep) {
        const char *p;
        size_t n;
        if (have_pending) {
            header.swap(pending);
            have_pending = false;
        } else {
            if (!lr.next(p, n)) return 0;
            if (n == 0 && lr.drained()) return 0;
            header.assign(p, n);
        }
        const size_t seq0 = b.seq.size();
        if (fmt == Fmt::Fasta) {
            if (header.empty() || header[0] != '>') { err = "FASTA: Expected > at record start."; return -1; }
            while (true) {
                const uint64_t at = lr.tell();
                if (!lr.next(p, n)) break;
                if (n && p[0] == '>') {
                    pending.assign(p, n);
                    have_pending = true;
                    pending_pos = at;
                    break;
                }
                n = trimmed_len(p, n);
                b.seq.insert(b.seq.end(), p, p + n);
            }
            b.off.push_back(b.seq.size());
            if (keep) {
                b.push_id(header.data(), header.size(), false);
                b.qual_off.push_back(b.qual.size());
                b.has_qual.push_back(0);
            }
            return 1;
        }
        if (header.empty() || header[0] != '@') { err = "FASTQ: Expected @ at record start."; return -1; }
        size_t lines_read = 0;
        while (lr.next(p, n)) {
            if (n && p[0] == '+') break;
            n = trimmed_len(p, n);
            b.seq.insert(b.seq.end(), p, p + n);
            ++lines_read;
        }
        const size_t q0 = b.qual.size();
        size_t qlen = 0;
        for (size_t i = 0; i < lines_read; ++i) {
            if (!lr.next(p, n)) break;  // read_line at end of file: nothing appended
            n = trimmed_len(p, n);
            if (keep) b.qual.insert(b.qual.end(), p, p + n);
            qlen += n;
        }
        if (qlen == 0) {
            b.seq.resize(seq0);
            b.qual.resize(q0);
            err = "FASTQ: Incomplete record.";
            return -1;
        }
        b.off.push_back(b.seq.size());
        if (keep) {
            b.push_id(header.data(), header.size(), true);
            b.qual_off.push_back(b.qual.size());
            b.has_qual.push_back(1);
        }
        return 1;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// ReadQueue (file_parser.rs:227-301): files consumed from the back of the list, records streamed across files.
//
// At 10^8 reads/s the text is the end-to-end limiter, so parsing is spread over `threads` workers:
//   * a plain file is memory-mapped and cut into chunks; every chunk is parsed on its own from the first record
//     start at or after its nominal begin to the first record start at or after its nominal end.  A FASTA record
//     start is any line beginning with '>' (exact); a FASTQ record start is guessed ('@' line followed by two
//     well-formed records) and then PROVEN by the consumer: chunk i+1 is accepted only if it starts exactly where
//     chunk i ended, which by induction from offset 0 makes every accepted start a true one.  A chunk that fails the
//     check is re-parsed from the proven position, so the result never depends on the guess;
//   * a gzip file is inflated and parsed by one worker as a stream of segments; several files run concurrently.
// Segments are consumed strictly in input order, so blocks, ids and outputs are those of a sequential reader.
// ---------------------------------------------------------------------------------------------------------------
struct Segment {
    Batch b;
    std::atomic<int> refs{0};             // the consumer + every batch that references the segment's ids / qualities
    std::vector<char> raw;                // plain chunks: the bytes of the file the records were parsed from
    uint64_t start_pos = 0, end_pos = 0;  // plain chunks: first record start, start of the record after the last one
    std::string err;                      // malformed input met after the records in b
    bool last = true;                     // gzip streams: more segments of this task follow when false
};
struct MappedFile {  // (plain files are read with pread into reused buffers: first-touch page faults of a mapping
    uint64_t size = 0;   //  cost more than the copy, and serialise the workers)
    int fd = -1;
    bool gz = false;
};
// bytes [lo, hi) of a file held in memory; p is the address byte 0 of the file would have
struct FileView {
    const char *p;
    uint64_t lo, hi;
};
struct Task {
    int file = 0;
    uint64_t lo = 0, hi = 0;  // nominal byte range of a plain chunk
    bool stream = false;      // gzip file: one streaming task
    std::deque<Segment *> out;
    bool taken = false;
};

struct ReadQueue {
    std::vector<std::string> files;   // consumption order (the reference pops from the back of its list)
    std::vector<Fmt> fmts;
    std::vector<MappedFile> maps;
    std::vector<Task> tasks;
    FmtOverride ov;
    bool keep = false;
    unsigned n_threads = 1;
    uint64_t chunk_bytes = 32ull << 20, seg_reads = 1u << 18;
    size_t lookahead = 8;

    std::mutex mu;
    std::condition_variable cv_work, cv_out;
    size_t next_task = 0, consume_task = 0;
    bool stop = false, started = false;
    std::vector<std::thread> workers;

    std::vector<Segment *> pool;  // consumed segments, handed back to the workers with their (warm) buffers
    Segment *cur = nullptr;  // segment being consumed
    size_t cur_read = 0;
    bool have_proven = false;
    uint64_t proven_pos = 0;  // plain files: where the next record of the current file provably starts

    ReadQueue(const std::string &path, FmtOverride o) : ov(o) {
        files = get_file_names(path);
        std::reverse(files.begin(), files.end());
        for (auto &f : files) fmts.push_back(detect_format(f, ov));
    }
    ~ReadQueue() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        cv_out.notify_all();
        for (auto &t : workers) t.join();
        delete cur;
        for (auto *s : pool) delete s;
        for (auto &t : tasks)
            for (auto *s : t.out) delete s;
        for (auto &m : maps)
            if (m.fd >= 0) close(m.fd);
    }
    Fmt peek_format() const { return files.empty() ? Fmt::Fasta : fmts.front(); }

    void start(bool keep_ids, unsigned threads) {
        keep = keep_ids;
        n_threads = std::max(1u, threads);
        if (!keep) {  // counts only: a segment is one device call, which wants >= 2^18 reads (bucketed path)
            chunk_bytes = 128ull << 20;
            seg_reads = 1u << 19;
        }
        if (const char *e = getenv("PFQ_INGEST_CHUNK_BYTES")) chunk_bytes = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
        if (const char *e = getenv("PFQ_INGEST_SEGMENT_READS")) seg_reads = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
        lookahead = (size_t)n_threads + 2;
        for (size_t i = 0; i < files.size(); ++i) {
            MappedFile m;
            m.fd = open(files[i].c_str(), O_RDONLY);
            if (m.fd < 0) die("Failed to open '" + files[i] + "': " + strerror(errno));
            struct stat st;
            if (fstat(m.fd, &st) != 0) die("cannot stat '" + files[i] + "'");
            m.size = (uint64_t)st.st_size;
            unsigned char magic[2] = {0, 0};
            m.gz = pread(m.fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
            if (!m.gz) posix_fadvise(m.fd, 0, 0, POSIX_FADV_SEQUENTIAL);
            maps.push_back(m);
            if (m.gz) {
                Task t;
                t.file = (int)i;
                t.stream = true;
                tasks.push_back(std::move(t));
            } else {
                for (uint64_t lo = 0; lo < std::max<uint64_t>(m.size, 1); lo += chunk_bytes) {
                    Task t;
                    t.file = (int)i;
                    t.lo = lo;
                    t.hi = std::min(m.size, lo + chunk_bytes);
                    tasks.push_back(std::move(t));
                }
            }
        }
        started = true;
        for (unsigned i = 0; i < n_threads; ++i) workers.emplace_back([this] { work(); });
    }

    // ---- chunk parsing ----------------------------------------------------------------------------------------
    // First record start at or after `from` (a line start) inside the view, or v.hi when there is none.
    uint64_t find_record_start(const FileView &v, Fmt fmt, uint64_t from) const {
        if (from == 0) return 0;
        uint64_t pos = from;
        if (v.p[pos - 1] != '\n') {  // not a line start: move to the next one
            const char *nl = (const char *)memchr(v.p + pos, '\n', v.hi - pos);
            if (!nl) return v.hi;
            pos = (uint64_t)(nl - v.p) + 1;
        }
        const char marker = fmt == Fmt::Fasta ? '>' : '@';
        while (pos < v.hi) {
            if (v.p[pos] == marker) {
                if (fmt == Fmt::Fasta) return pos;
                // FASTQ: '@' also starts quality lines; take the candidate if two records parse from it
                MemLines ml(v.p, pos, v.hi);
                RecordParser<MemLines> rp(ml, fmt);
                Batch scratch;
                int ok = rp.next(scratch, false);
                if (ok == 1) ok = rp.next(scratch, false);
                if (ok >= 0) return pos;  // (0: the file ends after the first record)
            }
            const char *nl = (const char *)memchr(v.p + pos, '\n', v.hi - pos);
            if (!nl) return v.hi;
            pos = (uint64_t)(nl - v.p) + 1;
        }
        return v.hi;
    }
    // Records starting in [start, nominal_end), parsed out of the view; returns whether the parser ran into the end of
    // the view (the caller reads more of the file and parses again unless the view ends where the file does).
    bool parse_view(const FileView &v, Fmt fmt, uint64_t start, uint64_t nominal_end, Segment &s) const {
        s.start_pos = start;
        if (nominal_end > start) {  // one allocation per buffer instead of a doubling series
            const uint64_t span = nominal_end - start;
            s.b.seq.reserve(span / (fmt == Fmt::Fastq ? 2 : 1) + 4096);
            s.b.off.reserve(span / 64 + 16);
        }
        MemLines ml(v.p, std::min(start, v.hi), v.hi);
        RecordParser<MemLines> rp(ml, fmt);
        while (rp.next_record_pos() < nominal_end) {
            int rc = rp.next(s.b, keep);
            if (rc == 0) break;
            if (rc < 0) {
                s.err = rp.err;
                break;
            }
        }
        s.end_pos = rp.next_record_pos();
        return ml.cur == ml.end;
    }
    // Chunk [lo, hi) of a plain file: records from the first record start at or after `lo` (or from `forced_start` when
    // the consumer knows it) up to the first record start at or after `hi`.
    void parse_chunk(const MappedFile &m, Fmt fmt, uint64_t lo, uint64_t hi, bool forced, uint64_t forced_start, Segment &s) {
        if (!m.size) return;
        const uint64_t want_lo = forced ? std::min(forced_start, m.size) : (lo ? lo - 1 : 0);
        for (uint64_t extra = 1ull << 20;; extra *= 8) {
            const uint64_t r_hi = std::min(m.size, std::max(hi, want_lo) + extra);
            s.raw.resize((size_t)(r_hi - want_lo));
            for (uint64_t got = 0; got < r_hi - want_lo;) {
                ssize_t n = pread(m.fd, s.raw.data() + got, (size_t)(r_hi - want_lo - got), (off_t)(want_lo + got));
                if (n < 0) die(std::string("read error: ") + strerror(errno));
                if (n == 0) die("input file shrank while it was being read");
                got += (uint64_t)n;
            }
            const FileView v{(const char *)((uintptr_t)s.raw.data() - (uintptr_t)want_lo), want_lo, r_hi};
            s.b.clear();
            s.err.clear();
            const uint64_t t0 = now_ns();
            const uint64_t start = forced ? forced_start : find_record_start(v, fmt, lo);
            const uint64_t t1 = now_ns();
            const bool hit_end = parse_view(v, fmt, start, hi, s);
            ns_find += t1 - t0;
            ns_parse += now_ns() - t1;
            if (!hit_end || r_hi == m.size) return;  // otherwise the last record may be cut: read further
        }
    }
    Segment *fresh_segment() {
        Segment *s = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!pool.empty()) {
                s = pool.back();
                pool.pop_back();
            }
        }
        if (!s) return new Segment;
        s->b.clear();
        s->start_pos = s->end_pos = 0;
        s->err.clear();
        s->last = true;
        return s;
    }
    void recycle(Segment *s) {
        std::lock_guard<std::mutex> lk(mu);
        if (pool.size() < lookahead + 4 + (keep ? 48 : 0)) pool.push_back(s);  // (filtering: batches in flight hold their segments)
        else delete s;
    }
    // one holder less; the last one hands the segment back (a segment that ends in malformed input is not reused)
    void release(Segment *s) {
        if (--s->refs > 0) return;
        if (!s->err.empty()) delete s;
        else recycle(s);
    }
    void release_held(Batch &b) {
        for (Segment *s : b.held) release(s);
        b.held.clear();
    }
    void push(Task &t, Segment *s) {
        {
            std::lock_guard<std::mutex> lk(mu);
            t.out.push_back(s);
        }
        cv_out.notify_all();
    }
    void work() {
        while (true) {
            size_t ti;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || (next_task < tasks.size() && next_task < consume_task + lookahead); });
                if (stop) return;
                ti = next_task++;
            }
            Task &t = tasks[ti];
            const MappedFile &m = maps[t.file];
            const Fmt fmt = fmts[t.file];
            if (!t.stream) {
                Segment *s = fresh_segment();
                parse_chunk(m, fmt, t.lo, t.hi, false, 0, *s);
                push(t, s);
                continue;
            }
            GzLines gl(files[t.file]);
            RecordParser<GzLines> rp(gl, fmt);
            while (true) {
                Segment *s = fresh_segment();
                int rc = 1;
                while (s->b.n() < seg_reads && (rc = rp.next(s->b, keep)) == 1) {}
                if (rc < 0) s->err = rp.err;
                s->last = rc != 1;
                {
                    std::unique_lock<std::mutex> lk(mu);  // bounded: at most four segments of a stream wait
                    cv_work.wait(lk, [&] { return stop || t.out.size() < 4; });
                    if (stop) {
                        delete s;
                        return;
                    }
                    t.out.push_back(s);
                }
                cv_out.notify_all();
                if (rc != 1) break;
            }
        }
    }

    // ---- ordered consumption ----------------------------------------------------------------------------------
    // Next segment in input order (validated), or nullptr at the end of the input.
    Segment *next_segment() {
        while (consume_task < tasks.size()) {
            Task &t = tasks[consume_task];
            Segment *s;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_out.wait(lk, [&] { return !t.out.empty(); });
                s = t.out.front();
                t.out.pop_front();
            }
            cv_work.notify_all();
            const bool task_done = s->last;
            if (!t.stream) {
                const MappedFile &m = maps[t.file];
                const uint64_t expect = t.lo == 0 ? 0 : proven_pos;
                if (m.size && s->start_pos != expect) {  // guessed boundary was wrong (or a record spans chunks): redo
                    parse_chunk(m, fmts[t.file], t.lo, t.hi, true, expect, *s);
                }
                proven_pos = s->end_pos;
            }
            if (task_done) {
                std::lock_guard<std::mutex> lk(mu);
                ++consume_task;
            }
            cv_work.notify_all();
            return s;
        }
        return nullptr;
    }
    // Appends up to max_reads reads (and at most ~max_bytes bases); false when the input is exhausted.
    // by_ref (needs keep): ids and qualities are referenced, not copied — the batch holds the segments (release_held).
    bool fill(Batch &b, uint64_t max_reads, uint64_t max_bytes, bool by_ref = false) {
        if (!started) die("ReadQueue::start was not called");
        while (b.n() < max_reads && b.seq.size() < max_bytes) {
            if (!cur) {
                const uint64_t t0 = now_ns();
                cur = next_segment();
                ns_wait += now_ns() - t0;
                cur_read = 0;
                if (!cur) return false;
                cur->refs = 1;  // this consumer
            }
            const size_t avail = cur->b.n() - cur_read;
            size_t take = (size_t)std::min<uint64_t>(avail, max_reads - b.n());
            if (take && max_bytes != ~0ull) {  // keep the byte bound (coarsely: per read)
                size_t t = 0;
                while (t < take && b.seq.size() + (cur->b.off[cur_read + t] - cur->b.off[cur_read]) < max_bytes) ++t;
                take = std::max<size_t>(t, 1);
            }
            const uint64_t t0 = now_ns();
            if (take && keep && by_ref) {
                b.append_ref(cur->b, cur_read, cur_read + take);
                if (b.held.empty() || b.held.back() != cur) {
                    b.held.push_back(cur);
                    ++cur->refs;
                }
            } else if (take) b.append(cur->b, cur_read, cur_read + take, keep);
            ns_append += now_ns() - t0;
            cur_read += take;
            if (cur_read == cur->b.n()) {
                if (!cur->err.empty()) {
                    pending_error = cur->err;
                    release(cur);
                    cur = nullptr;
                    return false;
                }
                release(cur);
                cur = nullptr;
            }
        }
        return true;
    }
    std::string pending_error;  // malformed input reached: fatal once the reads before it have been processed

    // PFQ_INGEST_TIMING=1: where the wall time of the reader went (stderr)
    std::atomic<uint64_t> ns_parse{0}, ns_find{0}, ns_wait{0}, ns_append{0};
    static uint64_t now_ns() {
        return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    void report_timing() const {
        if (!getenv("PFQ_INGEST_TIMING")) return;
        fprintf(stderr, "ingest: %u workers, parse %.3f s (sum over workers), boundary search %.3f s, consumer waited %.3f s, appended %.3f s\n",
                n_threads, ns_parse.load() * 1e-9, ns_find.load() * 1e-9, ns_wait.load() * 1e-9, ns_append.load() * 1e-9);
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
                             {"search-depth", 0, true}, {"pos-filter", 0, false}, {"neg-filter", 0, false}, {"format", 'F', true},
                             {"devices", 0, true}};
    Args a = parse(argc, argv, 2, opts);
    const std::string reads = req(a, "reads"), out = req(a, "out"), db = req(a, "db-path");
    const unsigned threads = (unsigned)std::min<uint64_t>(to_u64(opt(a, "threads", "4"), "threads"), 256);  // rayon pool size in the reference; here: parser workers
    (void)to_u64(opt(a, "cache-size", "10"), "cache-size");  // LRU of .bf files: the whole tree is resident in HBM
    const uint64_t block = to_u64(opt(a, "block-size-reads", "100"), "block-size-reads");
    const float threshold = to_f32(opt(a, "filter-threshold", "1.0"), "filter-threshold");
    const bool pos = a.flags.count("pos-filter") != 0, neg = a.flags.count("neg-filter") != 0;
    const bool filtering = pos || neg;
    const FmtOverride ov = to_fmt(opt(a, "format", "auto"));

    // --devices 0,1,..|all (or PFQ_DEVICES): one replica of the database per listed GPU, each fed by its own host thread;
    // the per-genome counts are combined by one RCCL all-reduce at the end.  The reference has one rayon pool instead
    // (main.rs:269-272); results do not depend on how the reads are dealt.
    std::vector<int> devices;
    {
        std::string spec = opt(a, "devices", getenv("PFQ_DEVICES") ? getenv("PFQ_DEVICES") : "");
        if (spec == "all") {
            int n = 0;
            check(pfq_device_count(&n));
            for (int i = 0; i < n; ++i) devices.push_back(i);
        } else if (!spec.empty()) {
            size_t p0 = 0;
            while (p0 <= spec.size()) {
                size_t p1 = spec.find(',', p0);
                if (p1 == std::string::npos) p1 = spec.size();
                devices.push_back((int)to_u64(spec.substr(p0, p1 - p0), "devices"));
                p0 = p1 + 1;
            }
        }
        if (devices.empty()) devices.push_back(device_from_env());
    }
    const size_t n_dev = devices.size();
    std::vector<pfq_tree *> trees(n_dev, nullptr);
    {   // BloomTree::load per replica, side by side
        std::vector<std::string> errs(n_dev);
        std::vector<std::thread> th;
        auto open_one = [&](size_t i) {
            if (pfq_tree_open(db.c_str(), devices[i], &trees[i]) != PFQ_OK) errs[i] = std::string("libpfq: ") + pfq_last_error();
        };
        for (size_t i = 1; i < n_dev; ++i) th.emplace_back(open_one, i);
        open_one(0);
        for (auto &t : th) t.join();
        for (auto &e : errs)
            if (!e.empty()) die(e);
    }
    pfq_tree *tree = trees[0];
    printf("Querying reads...\n");
    printf("Filtering settings: positive=%s; negative=%s\n", pos ? "true" : "false", neg ? "true" : "false");
    if (a.val.count("search-depth")) {
        uint64_t depth = to_u64(a.val.at("search-depth"), "search-depth");
        if (!filtering) printf("If using a search depth, use a filtering flag (--pos-filter or --neg-filter, or both!)\n");
        printf("Search depth settings: %llu\n", (unsigned long long)depth);
        for (pfq_tree *t : trees) check(pfq_tree_prune(t, depth));
    }
    ReadQueue rq(reads, ov);
    // Page-locking costs ~1.7 s per GB here (hipHostMalloc), the pageable copy ~0.1 s per GB: pinned buffers only
    // pay off once every pooled buffer has been reused a few dozen times (inputs of >~ 10^9 reads).  Opt-in.
    g_pinned = getenv("PFQ_PINNED") && atoi(getenv("PFQ_PINNED")) != 0;
    // --block-size-reads 0: the reference's first block is empty (file_parser.rs:252-270: `0 > read_block.len()` is false),
    // so its loop (main.rs:334-368) never runs: no read is parsed or classified, the outputs are created empty
    if (block != 0) rq.start(filtering, threads);

    // create_and_overwrite_directory (main.rs:380-391): an existing output directory is deleted
    struct stat st;
    if (stat(out.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) rm_rf(out);
    mkdir(out.c_str(), 0777);
    const char *ext = rq.peek_format() == Fmt::Fastq ? "fq" : "fa";
    int pos_fd = -1, neg_fd = -1;
    uint64_t pos_size = 0, neg_size = 0;  // bytes written so far (formatters write their parts at computed offsets)
    if (pos && (pos_fd = open((out + "/POS_FILTERING." + ext).c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666)) < 0)
        die("cannot create POS_FILTERING in " + out);
    if (neg && (neg_fd = open((out + "/NEG_FILTERING." + ext).c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666)) < 0)
        die("cannot create NEG_FILTERING in " + out);
    auto write_at = [](int fd, const char *buf, size_t len, uint64_t at) {
        for (size_t done = 0; done < len;) {
            ssize_t n = pwrite(fd, buf + done, len - done, (off_t)(at + done));
            if (n < 0) {
                fprintf(stderr, "phage_filter: write error: %s\n", strerror(errno));
                _exit(101);  // (called from writer threads: see fail_from_thread)
            }
            done += (size_t)n;
        }
    };

    const char *const *tax = nullptr;
    const uint64_t *cnt = nullptr;
    uint64_t n_leaves = 0;
    check(pfq_leaf_counts(tree, &tax, &cnt, &n_leaves));
    std::vector<std::string> leaf_names(tax, tax + n_leaves);

    const uint64_t t_loop0 = ReadQueue::now_ns();
    std::atomic<uint64_t> ns_gpu{0}, ns_out{0}, n_total{0};
    // An error on a worker thread ends the process at once, WITHOUT exit(): exit() would run the atexit handlers and static
    // destructors (the HIP runtime's among them) while the other replicas' threads, the parser and the formatters are still
    // inside HIP calls or writing.  Same message and status as die() (the reference panics: status 101).
    auto fail_from_thread = [](const char *what) {
        fprintf(stderr, "phage_filter: libpfq: %s\n", what);
        fflush(stderr);
        _exit(101);
    };
    if (block == 0) {
        // nothing to do: see above
    } else if (!filtering) {
        // Counts only: the result does not depend on how the reads are cut into device calls (mapped_reads just
        // accumulates, query.rs:143), so every parsed segment goes to a GPU as it is — no host-side copy.  Segments are
        // handed out in input order to whichever replica's thread asks next.
        std::mutex qm;
        bool done = false;
        auto device_loop = [&](size_t d) {
            while (true) {
                Segment *sg = nullptr;
                {
                    std::lock_guard<std::mutex> lk(qm);
                    if (done) return;
                    sg = rq.next_segment();
                    if (!sg) {
                        done = true;
                        return;
                    }
                    if (!sg->err.empty()) {  // malformed input: the reads before it are still classified, nothing after it
                        rq.pending_error = sg->err;
                        done = true;
                    }
                }
                const uint64_t n = sg->b.n();
                if (n) {
                    sg->b.seq.resize(sg->b.seq.size() + 16);
                    const uint64_t tq0 = ReadQueue::now_ns();
                    if (pfq_query_batch(trees[d], sg->b.seq.data(), sg->b.off.data(), n, threshold, 0, nullptr) != PFQ_OK)
                        fail_from_thread(pfq_last_error());
                    ns_gpu += ReadQueue::now_ns() - tq0;
                    n_total += n;
                }
                if (!sg->err.empty()) delete sg;
                else rq.recycle(sg);
            }
        };
        std::vector<std::thread> th;
        for (size_t d = 1; d < n_dev; ++d) th.emplace_back(device_loop, d);
        device_loop(0);
        for (auto &t : th) t.join();
    } else {
        // The device processes big batches; ResultMap semantics (ids merged per reference block, cleared per block,
        // main.rs:334-368) are applied per `block` consecutive reads so the outputs do not depend on the batch size.
        uint64_t batch_target = 1u << 20;
        if (const char *e = getenv("PFQ_CLI_BATCH_READS")) batch_target = std::max<uint64_t>(1, strtoull(e, nullptr, 10));  // (tests: several batches from a small input)
        const uint64_t batch_reads = std::max<uint64_t>(block, batch_target) / block * block;
        // Three stages, each on its own thread(s), batches in flight between them: (1) the assembler (above all the copy of
        // the parsed records into batches of whole reference blocks), (2) one thread per replica: pfq_query_batch with hits,
        // which are copied out of the library's buffers, (3) ONE output thread that takes the classified batches in input
        // order, formats them with `threads` workers and writes every worker's part at its offset (pwrite, side by side) —
        // while batch k is formatted and written, batch k + 1 is on the GPU and batch k + 2 is being assembled.
        const size_t NB = n_dev + 3;
        std::vector<Batch> batches(NB);
        std::vector<std::vector<uint64_t>> hit_off(NB);
        std::vector<std::vector<uint32_t>> hit_leaves(NB);
        std::vector<int> ready(NB, 0);          // 0 = free for the assembler, 1 = filled, 2 = classified
        std::vector<uint64_t> batch_seq(NB, 0);  // which batch a slot holds
        std::mutex mu;
        std::condition_variable cv;
        long long last_seq = -1;                 // sequence number of the last batch, once the assembler knows it
        uint64_t next_take = 0;
        std::atomic<uint64_t> ns_fmt{0}, ns_write{0};
        std::thread parser([&] {
            bool more = true;
            for (uint64_t k = 0; more; ++k) {
                const size_t slot = (size_t)(k % NB);
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return ready[slot] == 0; });
                }
                rq.release_held(batches[slot]);  // (written: its segments go back to the parsers)
                batches[slot].clear();
                more = rq.fill(batches[slot], batch_reads, 3ull << 30, true);
                {
                    std::lock_guard<std::mutex> lk(mu);
                    ready[slot] = 1;
                    batch_seq[slot] = k;
                    if (!more) last_seq = (long long)k;
                }
                cv.notify_all();
            }
        });
        auto device_loop = [&](size_t d) {
            while (true) {
                uint64_t k;
                size_t slot;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    k = next_take++;
                    slot = (size_t)(k % NB);
                    cv.wait(lk, [&] { return (ready[slot] == 1 && batch_seq[slot] == k) || (last_seq >= 0 && (long long)k > last_seq); });
                    if (last_seq >= 0 && (long long)k > last_seq) return;
                }
                Batch &b = batches[slot];
                const uint64_t n = b.n();
                hit_off[slot].assign(n + 1, 0);
                hit_leaves[slot].clear();
                if (n) {
                    b.seq.resize(b.seq.size() + 16);
                    pfq_hits hits{};
                    const uint64_t tq0 = ReadQueue::now_ns();
                    if (pfq_query_batch(trees[d], b.seq.data(), b.off.data(), n, threshold, PFQ_WANT_HITS, &hits) != PFQ_OK)
                        fail_from_thread(pfq_last_error());
                    // (library-owned until the next call on this replica: the output thread works on copies)
                    memcpy(hit_off[slot].data(), hits.offsets, (n + 1) * sizeof(uint64_t));
                    hit_leaves[slot].assign(hits.leaves, hits.leaves + hits.offsets[n]);
                    ns_gpu += ReadQueue::now_ns() - tq0;
                    n_total += n;
                }
                {
                    std::lock_guard<std::mutex> lk(mu);
                    ready[slot] = 2;
                }
                cv.notify_all();
            }
        };
        // Output buffers of the formatters: grown with realloc (no zero fill), kept from batch to batch.
        struct OutBuf {
            char *p = nullptr;
            size_t n = 0, cap = 0;
            ~OutBuf() { free(p); }
            char *room(size_t want) {
                if (n + want > cap) {
                    cap = std::max(n + want, cap + cap / 2 + (1u << 20));
                    p = (char *)realloc(p, cap);
                    if (!p) die("out of memory (output buffer)");
                }
                return p + n;
            }
            void put(const char *src, size_t len) {
                memcpy(room(len), src, len);
                n += len;
            }
            void put(char c) {
                *room(1) = c;
                ++n;
            }
        };
        // (formatters and writers share the cores with the parser workers and the assembler; measured on 16 cores with
        // -t 16: 16 formatters / 16 writers 18.6 - 19.2 M reads/s, 12 / 8: 19.5 - 19.7, 8 / 4: 16.8 - 17.1.  PFQ_CLI_FMT_WORKERS /
        // PFQ_CLI_WRITERS override the split.)
        unsigned fmt_workers = std::max(1u, threads * 3u / 4u);
        if (const char *e = getenv("PFQ_CLI_FMT_WORKERS")) fmt_workers = std::max(1u, (unsigned)atoi(e));
        unsigned max_writers = std::max(1u, threads / 2u);
        if (const char *e = getenv("PFQ_CLI_WRITERS")) max_writers = std::max(1u, (unsigned)atoi(e));
        // two sets of output buffers: while the writer thread puts set s into the files, the formatters fill the other one
        std::vector<OutBuf> pos_sets[2] = {std::vector<OutBuf>(fmt_workers), std::vector<OutBuf>(fmt_workers)};
        std::vector<OutBuf> neg_sets[2] = {std::vector<OutBuf>(fmt_workers), std::vector<OutBuf>(fmt_workers)};
        struct WriteJob {
            int state = 0;  // 0 free, 1 to be written
            unsigned nw = 0;
            std::vector<uint64_t> pos_at, neg_at;
        } jobs[2];
        bool writer_done = false;
        std::thread writer([&] {
            for (uint64_t j = 0;; ++j) {
                WriteJob &job = jobs[j & 1];
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return job.state == 1 || writer_done; });
                    if (job.state != 1) return;
                }
                const uint64_t t1 = ReadQueue::now_ns();
                std::vector<OutBuf> &pb = pos_sets[j & 1], &nb = neg_sets[j & 1];
                std::vector<std::thread> ts;
                const unsigned n_wr = std::min(job.nw, max_writers);
                auto write_parts = [&](unsigned x) {  // writer x takes parts x, x + n_wr, ...
                    for (unsigned w = x; w < job.nw; w += n_wr) {
                        if (pos_fd >= 0) write_at(pos_fd, pb[w].p, pb[w].n, job.pos_at[w]);
                        if (neg_fd >= 0) write_at(neg_fd, nb[w].p, nb[w].n, job.neg_at[w]);
                    }
                };
                for (unsigned x = 1; x < n_wr; ++x) ts.emplace_back(write_parts, x);
                write_parts(0);
                for (auto &t : ts) t.join();
                ns_write += ReadQueue::now_ns() - t1;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    job.state = 0;
                }
                cv.notify_all();
            }
        });
        std::thread output([&] {
            // ResultMap of one block (result_map.rs:20-45) without allocations: an open-addressing table over the ids of the
            // reads that hit.  An id that hits twice in a block (the same read id in two records) merges its genomes.
            // Two phases per batch: (1) workers take whole blocks: table of the block's hit ids, then for EVERY read of the
            // block its group (read_mapped: the id is in the block's map) — all hashing happens here; (2) workers take equal
            // ranges of reads and only format (a block of 100 000 reads is formatted by several workers).
            struct Group {
                std::string_view id;
                uint64_t first_read;          // the (first) read with this id that hit
                int32_t merged;               // index into the block's merged sets once a second read with the id has hit
            };
            std::vector<Group> groups;                            // [hit reads of the batch], a slice per block
            std::vector<uint32_t> table;                          // open-addressing tables of all blocks, a slice (power of two) per block
            std::vector<uint64_t> grp0, tab0;                     // [blocks + 1] first group / first table slot of every block
            std::vector<int32_t> grp_of;                          // [reads] group of the read within its block, -1: not mapped
            std::vector<std::vector<std::vector<uint32_t>>> merged_of;  // [blocks] merged genome sets (ids that hit more than once)
            auto hash_id = [](std::string_view v) {
                uint64_t h = 0x9E3779B97F4A7C15ull ^ v.size();
                size_t i = 0;
                for (; i + 8 <= v.size(); i += 8) {
                    uint64_t x;
                    memcpy(&x, v.data() + i, 8);
                    h = (h ^ x) * 0xff51afd7ed558ccdull;
                    h ^= h >> 32;
                }
                uint64_t x = 0;
                if (i < v.size()) memcpy(&x, v.data() + i, v.size() - i);
                h = (h ^ x) * 0xc4ceb9fe1a85ec53ull;
                return h ^ (h >> 29);
            };
            auto run_workers = [&](unsigned nw, const std::function<void(unsigned)> &fn) {
                std::vector<std::thread> ts;
                for (unsigned w = 1; w < nw; ++w) ts.emplace_back(fn, w);
                fn(0);
                for (auto &t : ts) t.join();
            };
            for (uint64_t k = 0;; ++k) {
                const size_t slot = (size_t)(k % NB);
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return (ready[slot] == 2 && batch_seq[slot] == k) || (last_seq >= 0 && (long long)k > last_seq); });
                    if (last_seq >= 0 && (long long)k > last_seq) return;
                }
                const Batch &b = batches[slot];
                const uint64_t n = b.n();
                const uint64_t *h_off = hit_off[slot].data();
                const uint32_t *h_leaves = hit_leaves[slot].data();
                {   // the buffer set of this batch must have been written
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return jobs[k & 1].state == 0; });
                }
                std::vector<OutBuf> &pos_buf = pos_sets[k & 1], &neg_buf = neg_sets[k & 1];
                const uint64_t t0 = ReadQueue::now_ns();
                const uint64_t n_blocks = n ? (n + block - 1) / block : 0;
                // ---- phase 1: the blocks' maps
                grp0.assign(n_blocks + 1, 0);
                tab0.assign(n_blocks + 1, 0);
                for (uint64_t blk = 0; blk < n_blocks; ++blk) {
                    const uint64_t b0 = blk * block, b1 = std::min(n, b0 + block);
                    uint64_t n_hit = 0;
                    for (uint64_t r = b0; r < b1; ++r) n_hit += h_off[r] != h_off[r + 1];
                    uint64_t want = 0;
                    if (n_hit) {
                        want = 16;
                        while (want < 2 * n_hit) want <<= 1;
                    }
                    grp0[blk + 1] = grp0[blk] + n_hit;
                    tab0[blk + 1] = tab0[blk] + want;
                }
                if (groups.size() < grp0[n_blocks]) groups.resize(grp0[n_blocks]);
                table.assign(tab0[n_blocks], 0);
                if (grp_of.size() < n) grp_of.resize(n);
                merged_of.resize(std::max<size_t>(merged_of.size(), n_blocks));
                const unsigned nw1 = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(fmt_workers, n_blocks));
                run_workers(nw1, [&](unsigned w) {
                    for (uint64_t blk = n_blocks * w / nw1; blk < n_blocks * (w + 1) / nw1; ++blk) {
                        const uint64_t b0 = blk * block, b1 = std::min(n, b0 + block);
                        Group *grp = groups.data() + grp0[blk];
                        uint32_t *tab = table.data() + tab0[blk];
                        const size_t mask = (size_t)(tab0[blk + 1] - tab0[blk]) - 1;   // (no table: the block has no hit)
                        const bool any = tab0[blk + 1] != tab0[blk];
                        auto &merged = merged_of[blk];
                        merged.clear();
                        uint32_t n_grp = 0;
                        auto find = [&](std::string_view id, bool insert, uint64_t r) -> int32_t {
                            for (size_t i = (size_t)hash_id(id) & mask;; i = (i + 1) & mask) {
                                const uint32_t g = tab[i];
                                if (!g) {
                                    if (!insert) return -1;
                                    grp[n_grp] = Group{id, r, -1};
                                    tab[i] = ++n_grp;
                                    return (int32_t)n_grp - 1;
                                }
                                if (grp[g - 1].id == id) return (int32_t)g - 1;
                            }
                        };
                        if (any)
                            for (uint64_t r = b0; r < b1; ++r) {
                                if (h_off[r] == h_off[r + 1]) continue;
                                const int32_t g = find(b.id(r), true, r);
                                grp_of[r] = g;
                                Group &gr = grp[g];
                                if (gr.first_read == r) continue;        // new id
                                if (gr.merged < 0) {                      // second read with this id: the sets merge
                                    gr.merged = (int32_t)merged.size();
                                    merged.emplace_back(h_leaves + h_off[gr.first_read], h_leaves + h_off[gr.first_read + 1]);
                                }
                                merged[gr.merged].insert(merged[gr.merged].end(), h_leaves + h_off[r], h_leaves + h_off[r + 1]);
                            }
                        for (auto &v : merged) {  // a set of genomes per id; printed in leaf order
                            std::sort(v.begin(), v.end());
                            v.erase(std::unique(v.begin(), v.end()), v.end());
                        }
                        for (uint64_t r = b0; r < b1; ++r)   // read_mapped: the id is in the block's map (also for reads that did not hit themselves)
                            if (h_off[r] == h_off[r + 1]) grp_of[r] = any ? find(b.id(r), false, 0) : -1;
                    }
                });
                // ---- phase 2: the records
                const unsigned nw = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(fmt_workers, (n + 4095) / 4096));
                run_workers(nw, [&](unsigned w) {
                    OutBuf &pb = pos_buf[w], &nb = neg_buf[w];
                    pb.n = nb.n = 0;
                    for (uint64_t r = n * w / nw; r < n * (w + 1) / nw; ++r) {
                        const int32_t g = grp_of[r];
                        const bool mapped = g >= 0;
                        if (mapped ? !pos : !neg) continue;
                        OutBuf &line = mapped ? pb : nb;
                        const std::string_view id = b.id(r);
                        const uint64_t len = b.off[r + 1] - b.off[r];
                        const bool fq = b.has_qual[r] != 0;
                        {   // write_record (main.rs:394-404): '@' / '>' + id
                            char *dst = line.room(id.size() + 3);
                            dst[0] = fq ? '@' : '>';
                            memcpy(dst + 1, id.data(), id.size());
                            line.n += id.size() + 1;
                        }
                        if (mapped) {                              // get_ext_id: "{id} |{g1,g2}" (set order unspecified in the reference)
                            line.put(" |", 2);
                            const uint64_t blk = r / block;
                            const Group &gr = groups[grp0[blk] + (uint64_t)g];
                            const uint32_t *l0, *l1;
                            if (gr.merged >= 0) {
                                l0 = merged_of[blk][gr.merged].data();
                                l1 = l0 + merged_of[blk][gr.merged].size();
                            } else {
                                l0 = h_leaves + h_off[gr.first_read];   // (ascending within a read, no duplicates)
                                l1 = h_leaves + h_off[gr.first_read + 1];
                            }
                            for (const uint32_t *l = l0; l < l1; ++l) {
                                if (l != l0) line.put(',');
                                line.put(leaf_names[*l].data(), leaf_names[*l].size());
                            }
                        }
                        const std::string_view q = fq ? b.quality(r) : std::string_view();
                        char *dst = line.room(len + q.size() + 5);
                        *dst++ = '\n';
                        copy_upper(dst, b.seq.data() + b.off[r], len);   // the sequence, upper-cased (main.rs:347-349)
                        dst += len;
                        *dst++ = '\n';
                        size_t wrote = len + 2;
                        if (fq) {
                            dst[0] = '+';
                            dst[1] = '\n';
                            memcpy(dst + 2, q.data(), q.size());
                            dst[2 + q.size()] = '\n';
                            wrote += q.size() + 3;
                        }
                        line.n += wrote;
                    }
                });
                for (unsigned w = nw; w < fmt_workers; ++w) pos_buf[w].n = neg_buf[w].n = 0;
                const uint64_t t1 = ReadQueue::now_ns();
                ns_fmt += t1 - t0;
                // the bytes: batches come in input order, so a part's place is the sum of what lies before it
                std::vector<uint64_t> pos_at(nw + 1, pos_size), neg_at(nw + 1, neg_size);
                for (unsigned x = 0; x < nw; ++x) {
                    pos_at[x + 1] = pos_at[x] + pos_buf[x].n;
                    neg_at[x + 1] = neg_at[x] + neg_buf[x].n;
                }
                pos_size = pos_at[nw];
                neg_size = neg_at[nw];
                ns_out += ReadQueue::now_ns() - t0;
                {   // the batch is free again (everything it holds is in the buffers); the writer takes over
                    std::lock_guard<std::mutex> lk(mu);
                    ready[slot] = 0;
                    jobs[k & 1].nw = nw;
                    jobs[k & 1].pos_at = pos_at;
                    jobs[k & 1].neg_at = neg_at;
                    jobs[k & 1].state = 1;
                }
                cv.notify_all();
            }
        });
        std::vector<std::thread> th;
        for (size_t d = 1; d < n_dev; ++d) th.emplace_back(device_loop, d);
        device_loop(0);
        for (auto &t : th) t.join();
        parser.join();
        output.join();
        {
            std::lock_guard<std::mutex> lk(mu);
            writer_done = true;
        }
        cv.notify_all();
        writer.join();
        for (Batch &bb : batches) rq.release_held(bb);
        if (getenv("PFQ_INGEST_TIMING"))
            fprintf(stderr, "output: format %.3f s, write %.3f s (%u workers; wall times of the formatter and the writer thread, which overlap)\n",
                    ns_fmt.load() * 1e-9, ns_write.load() * 1e-9, fmt_workers);
    }
    if (getenv("PFQ_INGEST_TIMING")) {
        const double wall = (ReadQueue::now_ns() - t_loop0) * 1e-9;
        fprintf(stderr, "query loop: %llu reads in %.3f s = %.2f M reads/s on %zu device(s) (pfq_query_batch %.3f s, output %.3f s)\n",
                (unsigned long long)n_total.load(), wall, n_total.load() / wall * 1e-6, n_dev, ns_gpu.load() * 1e-9, ns_out.load() * 1e-9);
        rq.report_timing();
        struct rusage ru;
        if (getrusage(RUSAGE_SELF, &ru) == 0)
            fprintf(stderr, "cpu: user %.2f s + system %.2f s so far (all threads) for %.2f s of query loop\n",
                    ru.ru_utime.tv_sec + ru.ru_utime.tv_usec * 1e-6, ru.ru_stime.tv_sec + ru.ru_stime.tv_usec * 1e-6, wall);
    }
    if (pos_fd >= 0) close(pos_fd);
    if (neg_fd >= 0) close(neg_fd);
    if (!rq.pending_error.empty()) die(rq.pending_error);  // the reads before the malformed record were processed
    // per-genome counts of all replicas: one RCCL all-reduce (every replica then holds the totals); replica 0 writes the file
    if (n_dev > 1) check(pfq_trees_allreduce_counts(trees.data(), (uint32_t)n_dev));
    check(pfq_save_leaf_counts(tree, (out + "/CLASSIFICATION.csv").c_str()));
    for (pfq_tree *t : trees) pfq_tree_close(t);
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
    rq.start(true, (unsigned)std::min<uint64_t>(to_u64(opt(a, "threads", "4"), "threads"), 256));
    Batch g;
    while (rq.fill(g, ~0ull, ~0ull)) {}  // block size 1 in the reference: one leaf per record (main.rs:148-200)
    if (!rq.pending_error.empty()) die(rq.pending_error);
    auto &seq = g.seq;
    auto &off = g.off;
    std::vector<std::string> ids;
    for (size_t r = 0; r < g.n(); ++r) ids.emplace_back(g.id(r));
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

// ---------------------------------------------------------------------------------------------------------------
// build / add (main.rs:148-247): one leaf per record, greedy placement by Hamming distance on the device
// ---------------------------------------------------------------------------------------------------------------
uint64_t random_seed() {  // HashSeed::new (hasher.rs:24-30) draws a random usize
    std::random_device rd;
    return ((uint64_t)rd() << 32) ^ (uint64_t)rd();
}
int insert_genomes(pfq_tree *tree, const std::string &genomes, FmtOverride ov, unsigned threads) {
    ReadQueue rq(genomes, ov);
    rq.start(true, threads);
    // the reference reads blocks of one record (ReadQueue::with_format(genomes, 1, …), main.rs:172,:229) and inserts
    // them in input order
    uint64_t n = 0;
    while (Segment *sg = rq.next_segment()) {
        for (size_t r = 0; r < sg->b.n(); ++r, ++n) {
            const std::string id(sg->b.id(r));
            check(pfq_tree_insert(tree, sg->b.seq.data() + sg->b.off[r], sg->b.off[r + 1] - sg->b.off[r], id.c_str(), nullptr));
        }
        if (!sg->err.empty()) die(sg->err);
        rq.recycle(sg);
    }
    return (int)n;
}
int cmd_build(int argc, char **argv) {
    std::vector<Opt> opts = {{"genomes", 'g', true}, {"db-path", 'd', true}, {"threads", 't', true}, {"kmer-size", 'k', true},
                             {"cache-size", 'c', true}, {"false-pos-rate", 'f', true}, {"largest-genome", 'l', true},
                             {"format", 'F', true}, {"seed1", 0, true}, {"seed2", 0, true}};
    Args a = parse(argc, argv, 2, opts);
    const std::string genomes = req(a, "genomes"), db = req(a, "db-path");
    const uint64_t k = to_u64(opt(a, "kmer-size", "20"), "kmer-size");
    const float fpr = to_f32(opt(a, "false-pos-rate", "0.001"), "false-pos-rate");
    const uint32_t largest = (uint32_t)to_u64(opt(a, "largest-genome", "1000000"), "largest-genome");
    (void)to_u64(opt(a, "cache-size", "10"), "cache-size");  // every filter stays in HBM while the tree is built
    const unsigned threads = (unsigned)std::min<uint64_t>(to_u64(opt(a, "threads", "4"), "threads"), 256);
    // --seed1/--seed2 are ours: the reference always draws the two hash seeds at random (bloom_tree.rs:114)
    const uint64_t s1 = a.val.count("seed1") ? strtoull(a.val.at("seed1").c_str(), nullptr, 0) : random_seed();
    const uint64_t s2 = a.val.count("seed2") ? strtoull(a.val.at("seed2").c_str(), nullptr, 0) : random_seed();
    printf("Building the SBT...\n");
    mkdir(db.c_str(), 0777);  // BloomTree::new creates the directory (bloom_tree.rs:107)
    pfq_tree *tree = nullptr;
    check(pfq_tree_create(k, fpr, largest, s1, s2, 0, device_from_env(), &tree));
    insert_genomes(tree, genomes, to_fmt(opt(a, "format", "auto")), threads);
    check(pfq_tree_save(tree, db.c_str()));
    pfq_tree_close(tree);
    printf("Finished.\n");
    return 0;
}
int cmd_add(int argc, char **argv) {
    std::vector<Opt> opts = {{"genomes", 'g', true}, {"db-path", 'd', true}, {"threads", 't', true}, {"cache-size", 'c', true},
                             {"format", 'F', true}};
    Args a = parse(argc, argv, 2, opts);
    const std::string genomes = req(a, "genomes"), db = req(a, "db-path");
    (void)to_u64(opt(a, "cache-size", "10"), "cache-size");
    const unsigned threads = (unsigned)std::min<uint64_t>(to_u64(opt(a, "threads", "4"), "threads"), 256);
    printf("Adding new genomes to the SBT...\n");
    pfq_tree *tree = nullptr;
    check(pfq_tree_open(db.c_str(), device_from_env(), &tree));
    insert_genomes(tree, genomes, to_fmt(opt(a, "format", "auto")), threads);
    check(pfq_tree_save(tree, db.c_str()));
    pfq_tree_close(tree);
    printf("Finished.\n");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// ingest-check: parse the input exactly like `query` does and print what was read (no GPU involved).  Used by the
// CPU tests to pin the parallel reader against a sequential one and against the reference's parsing rules.
// ---------------------------------------------------------------------------------------------------------------
int cmd_ingest_check(int argc, char **argv) {
    std::vector<Opt> opts = {{"reads", 'r', true}, {"threads", 't', true}, {"format", 'F', true}, {"dump", 0, false},
                             {"count", 0, false}, {"block-size-reads", 'b', true}};
    Args a = parse(argc, argv, 2, opts);
    ReadQueue rq(req(a, "reads"), to_fmt(opt(a, "format", "auto")));
    const bool count_only = a.flags.count("count") != 0;  // what `query` without filtering keeps: bases and offsets only
    rq.start(!count_only, (unsigned)std::min<uint64_t>(to_u64(opt(a, "threads", "4"), "threads"), 256));
    const uint64_t block = std::max<uint64_t>(1, to_u64(opt(a, "block-size-reads", "1000000"), "block-size-reads"));
    const bool dump = a.flags.count("dump") != 0;
    uint64_t n = 0, bytes = 0, h = 0xcbf29ce484222325ull;
    auto mix = [&](const void *p, size_t len) {
        const unsigned char *c = (const unsigned char *)p;
        for (size_t i = 0; i < len; ++i) h = (h ^ c[i]) * 0x100000001b3ull;
        h = (h ^ 0xff) * 0x100000001b3ull;
    };
    Batch b;
    bool more = true;
    while (more) {
        b.clear();
        more = rq.fill(b, block, ~0ull);
        if (count_only) {
            n += b.n();
            bytes += b.seq.size();
            continue;
        }
        for (size_t r = 0; r < b.n(); ++r) {
            const std::string_view id = b.id(r);
            const uint64_t o = b.off[r], len = b.off[r + 1] - o;
            mix(id.data(), id.size());
            mix(b.seq.data() + o, len);
            const std::string_view q = b.quality(r);
            if (b.has_qual[r]) mix(q.data(), q.size());
            if (dump) {
                printf("%c%.*s\x01%.*s\x01%.*s\n", b.has_qual[r] ? '@' : '>', (int)id.size(), id.data(), (int)len,
                       (const char *)b.seq.data() + o, (int)q.size(), q.data());
            }
            ++n;
            bytes += len;
        }
    }
    printf("reads=%llu bases=%llu fnv=%016llx\n", (unsigned long long)n, (unsigned long long)bytes, (unsigned long long)h);
    rq.report_timing();
    if (!rq.pending_error.empty()) die(rq.pending_error);
    return 0;
}

void usage() {
    fprintf(stderr,
            "A fast, simple and memory efficient metagenomic filtering tool. (MI355X query path)\n\n"
            "Usage: phage_filter [-v...|-q...] <COMMAND>\n\nCommands:\n"
            "  query           Queries a set of reads. (ran after building the bloom tree)\n"
            "  build           Builds the BloomTree.\n"
            "  add             Adds genomes to an already built BloomFilter.\n"
            "  build-balanced  Builds a balanced synthetic BloomTree on the GPU (benchmark databases)\n"
            "  ingest-check    Parses reads like `query` and prints what was read (no GPU)\n\n"
            "query takes the reference's options, plus --devices <0,1,..|all>: one replica of the database per GPU, reads\n"
            "dealt over them, per-genome counts combined by one RCCL all-reduce (default: device $PFQ_DEVICE or 0)\n");
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
    if (cmd == "ingest-check") return cmd_ingest_check((int)av.size(), av.data());
    if (cmd == "build") return cmd_build((int)av.size(), av.data());
    if (cmd == "add") return cmd_add((int)av.size(), av.data());
    usage();
    return 2;
}
