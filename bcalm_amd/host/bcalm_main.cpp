// bcalm -- command-line host of the MI355X-native constructor.
//
// Keeps the reference's CLI surface and output contract, re-stated from scratch:
//   ./bcalm -in reads.fa -kmer-size 31 -abundance-min 2 [-out prefix] [-minimizer-size m]
//     /root/reference/README.md:11-25 (usage), src/bcalm_1.cpp:55,61 (-in required,
//     "Specifiy -in"), :68-74 (<prefix> = -out or basename of -in), src/main.cpp:30-37
//     (-version / -v), :39-48 (error -> "EXCEPTION: ..." on stdout, exit code 1),
//     README.md:45-50 (FASTA/FASTQ, gzipped or not, or a file listing input files),
//     README.md:62-72 (>id LN:i: KC:i: km:f: header), scripts/convertToGFA.py:74,36-49 (GFA).
// Everything between parsing and writing is three calls into libcdbg.so (include/cdbg.h):
// this file is the replacement for bcalm_1::execute()/Functor (src/bcalm_1.cpp:49-97).
#include <zlib.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <algorithm>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "cdbg.h"
#include "pgz.h"

#ifndef CDBG_VERSION
#define CDBG_VERSION "cdbg-mi355x r1 (CLI-compatible with BCALM 2 v2.2.3)"
#endif

namespace {

struct Options {
    std::string in, out;
    int k = 31, amin = 2, m = 0, device = 0, log_np = -1, n_gpus = 1, cores = 0;
    bool gfa = false, verbose = false, all_ab = false, no_stream = false;
    std::string solid_out;
};

[[noreturn]] void usage_error(const std::string& msg) { throw std::runtime_error(msg); }

Options parse(int argc, char** argv) {
    Options o;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto need = [&](const char* name) -> const char* {
            if (i + 1 >= argc) usage_error(std::string("option ") + name + " needs a value");
            return argv[++i];
        };
        if (a == "-in") o.in = need("-in");
        else if (a == "-out") o.out = need("-out");
        else if (a == "-kmer-size") o.k = atoi(need("-kmer-size"));
        else if (a == "-abundance-min") o.amin = atoi(need("-abundance-min"));
        else if (a == "-minimizer-size") o.m = atoi(need("-minimizer-size"));
        else if (a == "-device") o.device = atoi(need("-device"));
        else if (a == "-log2-partitions") o.log_np = atoi(need("-log2-partitions"));
        else if (a == "-nb-gpus") o.n_gpus = atoi(need("-nb-gpus"));   // the GPU path's counterpart of -nb-cores: GPUs of this node (power of two)
        else if (a == "-gfa") o.gfa = true;
        else if (a == "-no-stream-scan") o.no_stream = true;           // dev: do not announce the input volume (the read scan starts when the text is complete)
        else if (a == "-all-abundance-counts") o.all_ab = true;        // README.md:74-80
        else if (a == "-solid-kmers-out") o.solid_out = need("-solid-kmers-out");   // hidden in the reference (bcalm_1.cpp:37)
        else if (a == "-verbose") { o.verbose = true; if (i + 1 < argc && argv[i + 1][0] != '-') ++i; }
        else if (a == "-nb-cores") o.cores = atoi(need("-nb-cores"));   // README.md / gatb option: here the host threads that parse the input and format the output (0 = all, at most 32)
        else if (a == "-max-memory" || a == "-max-disk" || a == "-out-tmp" || a == "-out-dir" ||
                 a == "-repartition-type" || a == "-minimizer-type" || a == "-histo-max" || a == "-solidity-kind")
            need(a.c_str());                     // accepted for CLI compatibility; meaningless on the GPU path
        else usage_error("Unknown parameter '" + a + "'");
    }
    return o;
}

std::string base_name(const std::string& path) {          // strip directory and the last extension
    size_t s = path.find_last_of('/');
    std::string b = s == std::string::npos ? path : path.substr(s + 1);
    size_t d = b.find_last_of('.');
    if (d != std::string::npos && d > 0) b = b.substr(0, d);
    if (b.size() > 3 && b.substr(b.size() - 3) == ".fa") b = b.substr(0, b.size() - 3);   // reads.fa.gz -> reads
    return b;
}

// ---- ingest (README.md:45-50: FASTA / FASTQ, gzipped or not, or a file listing input files) ----
// N parser threads write sequence text STRAIGHT INTO the library's pinned staging buffers (cdbg_stage_acquire / _commit: no
// intermediate copy; the H2D copies and -- with cdbg_expect_input -- the read scan itself run behind the parsing).  A plain
// file is memory-mapped and cut into record-aligned slices (FASTA: at "\n>", strict four-line FASTQ: at an '@' line whose
// second successor starts with '+'), one task per slice; a FASTQ whose records wrap is one task.  A gzip file is inflated by ALL
// threads when there are few of them (pgz.h: block starts searched, chunks decoded with the window unknown, resolved in order) and
// its text parsed in slices like a plain file; a list of many gzip files keeps one zlib stream per thread in flight instead.
struct Ingest {
    std::vector<cdbg_ctx*> ctxs; int k = 31;
    std::atomic<size_t> next_ctx{0};
    std::atomic<uint64_t> n_seq{0}, n_bases{0};
    std::mutex err_mu; std::string error; std::atomic<bool> failed{false}, irregular{false};
    void fail(const std::string& e) { std::lock_guard<std::mutex> l(err_mu); if (error.empty()) error = e; failed = true; }
};
// one per parser thread: the staging buffer being filled.  A sequence that does not fit is continued in the next buffer
// with its last k-1 bases repeated (every k-mer across the cut is seen exactly once).
class Sink {
    Ingest& I; cdbg_ctx* ctx = nullptr; char* buf = nullptr; uint64_t cap = 0, fill = 0, cur = 0;   // cur: bases of the open sequence in this buffer
    void acquire() {
        ctx = I.ctxs[I.next_ctx.fetch_add(1) % I.ctxs.size()];      // (several GPUs: whole buffers of complete sequences go to the contexts in turn)
        if (cdbg_stage_acquire(ctx, &buf, &cap) != 0) { buf = nullptr; throw std::runtime_error(cdbg_last_error()); }
        fill = 0; cur = 0;
    }
    void commit() {
        if (!buf) return;
        char* b = buf; buf = nullptr;
        if (cdbg_stage_commit(ctx, b, fill) != 0) throw std::runtime_error(cdbg_last_error());
    }
public:
    uint64_t n_seq = 0, n_bases = 0;
    explicit Sink(Ingest& i) : I(i) {}
    void bases(const char* p, size_t n) {
        n_bases += n;
        while (n) {
            if (!buf) acquire();
            const uint64_t room = cap - 1 - fill;                  // (the last byte is kept for the separator)
            if (!room) {
                const uint64_t t = std::min<uint64_t>(cur, (uint64_t)I.k - 1);
                std::string tail(buf + fill - t, t);
                if (cur < (uint64_t)I.k) fill -= cur;              // (fewer than k bases of it here: the whole piece moves on)
                buf[fill++] = '\n';
                commit(); acquire();
                memcpy(buf, tail.data(), t); fill = t; cur = t;
                continue;
            }
            const size_t take = (size_t)std::min<uint64_t>(room, n);
            memcpy(buf + fill, p, take); fill += take; cur += take; p += take; n -= take;
        }
    }
    void end_seq() {                                               // (cap - 1 >= fill always: there is room)
        if (buf && fill && buf[fill - 1] != '\n') buf[fill++] = '\n';
        cur = 0;
        if (buf && cap - fill < (1u << 16)) commit();               // nearly full: send it while the boundary is clean
    }
    void finish() { end_seq(); if (buf) commit(); I.n_seq += n_seq; I.n_bases += n_bases; n_seq = n_bases = 0; }
    ~Sink() { if (buf) cdbg_stage_commit(ctx, buf, 0); }           // (error path: hand the buffer back unused)
};
inline size_t line_len(const char* p, const char* end) { const char* e = (const char*)memchr(p, '\n', (size_t)(end - p)); return e ? (size_t)(e - p) : (size_t)(end - p); }
inline size_t rstrip_cr(const char* p, size_t n) { while (n && p[n - 1] == '\r') --n; return n; }

// a record-aligned slice of a memory-mapped plain file: FASTA (wrapped or not) ...
void parse_fasta_slice(const char* p, const char* end, Sink& out) {
    while (p < end) {
        const size_t n = line_len(p, end), m = rstrip_cr(p, n);
        if (m && p[0] == '>') { out.end_seq(); ++out.n_seq; }
        else if (m && p[0] != ';') out.bases(p, m);
        p += n + 1;
    }
    out.end_seq();
}
// ... or strict four-line FASTQ; returns false when a record is not of that shape (the caller falls back to the tolerant serial parser)
bool parse_fastq4_slice(const char* p, const char* end, Sink& out) {
    while (p < end) {
        const size_t n0 = line_len(p, end);
        if (!rstrip_cr(p, n0)) { p += n0 + 1; continue; }          // blank line between records
        const char* s = p + n0 + 1; if (s >= end || p[0] != '@') return false;
        const size_t n1 = line_len(s, end);
        const char* pl = s + n1 + 1; if (pl >= end || pl[0] != '+') return false;
        const size_t n2 = line_len(pl, end);
        const char* q = pl + n2 + 1; if (q > end) return false;
        const size_t n3 = q < end ? line_len(q, end) : 0;
        const size_t m1 = rstrip_cr(s, n1);
        if (rstrip_cr(q, n3) != m1) return false;
        out.bases(s, m1); out.end_seq(); ++out.n_seq;
        p = q + n3 + 1;
    }
    return true;
}
// the tolerant serial parser: FASTA / FASTQ, plain or gzip (zlib reads both), FASTQ records may wrap their sequence and quality
// over several lines (README.md:45-50 accepts any FASTQ): after the '@' header the sequence runs until the '+' line, the
// quality until it is as long as the sequence
void parse_stream(const std::string& path, Sink& out) {
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) usage_error("cannot open input file " + path);
    gzbuffer(f, 1 << 20);
    std::vector<char> line(1 << 22);
    int fmt = 0;                                // 0 unknown, 1 FASTA, 2 FASTQ
    int fq_state = 0; char fq_kind = 0; uint64_t fq_seq = 0, fq_qual = 0;   // state: 0 header expected, 1 sequence, 2 quality
    bool partial = false;                       // previous gzgets returned an unterminated piece
    while (gzgets(f, line.data(), (int)line.size())) {
        size_t n = strlen(line.data());
        const bool complete = n && line[n - 1] == '\n';
        while (n && (line[n - 1] == '\n' || line[n - 1] == '\r')) --n;
        const bool starts_line = !partial;
        partial = !complete;
        if (fmt == 0 && starts_line && n) fmt = line[0] == '@' ? 2 : 1;
        if (fmt == 2) {
            if (starts_line) {
                if (fq_state == 0) {
                    if (!n) fq_kind = 0;
                    else if (line[0] == '@') fq_kind = 'H';
                    else { gzclose(f); usage_error("malformed FASTQ record in " + path + " (sequence " + std::to_string(out.n_seq + 1) + "): '@' expected"); }
                } else if (fq_state == 1) fq_kind = (n && line[0] == '+') ? 'P' : 'S';
                else fq_kind = 'Q';
            }
            if (fq_kind == 'S') { out.bases(line.data(), n); fq_seq += n; }
            else if (fq_kind == 'Q') fq_qual += n;
            if (complete) {
                if (fq_kind == 'H') { fq_state = 1; fq_seq = 0; }
                else if (fq_kind == 'P') { out.end_seq(); ++out.n_seq; fq_state = 2; fq_qual = 0; }
                else if (fq_kind == 'Q' && fq_qual >= fq_seq) fq_state = 0;
            }
        } else {
            if (starts_line && n && line[0] == '>') { out.end_seq(); ++out.n_seq; }
            else if (!(starts_line && n && line[0] == ';')) out.bases(line.data(), n);
        }
    }
    int zerr = Z_OK; const char* zmsg = gzerror(f, &zerr);
    if (zerr != Z_OK && zerr != Z_STREAM_END) { const std::string m = zmsg ? zmsg : "read error"; gzclose(f); usage_error(path + ": " + m); }
    gzclose(f);
    out.end_seq();
}

struct Mapped {                                  // a plain file, memory-mapped read-only
    const char* p = nullptr; size_t n = 0; int fd = -1;
    bool open(const std::string& path) {
        fd = ::open(path.c_str(), O_RDONLY); if (fd < 0) return false;
        struct stat st; if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size == 0) { ::close(fd); fd = -1; return false; }
        n = (size_t)st.st_size;
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { ::close(fd); fd = -1; return false; }
        p = (const char*)m; (void)madvise(m, n, MADV_WILLNEED);
        return true;
    }
    ~Mapped() { if (p) munmap((void*)p, n); if (fd >= 0) ::close(fd); }
};
struct Task { int kind; std::string path; std::shared_ptr<Mapped> map; size_t beg = 0, end = 0; };   // kind: 0 serial stream, 1 FASTA slice, 2 four-line FASTQ slice
bool is_gzip(const std::string& path) {
    FILE* raw = fopen(path.c_str(), "rb"); unsigned char magic[2] = {0, 0};
    if (raw) { if (fread(magic, 1, 2, raw) != 2) magic[0] = 0; fclose(raw); }
    return magic[0] == 0x1f && magic[1] == 0x8b;
}
// start of the first record at or after `pos`
size_t next_fasta_record(const char* p, size_t n, size_t pos) {
    if (pos == 0) return 0;
    for (size_t i = pos - 1; i + 1 < n;) {
        const char* e = (const char*)memchr(p + i, '\n', n - i);
        if (!e) return n;
        i = (size_t)(e - p) + 1;
        if (i < n && p[i] == '>') return i;
    }
    return n;
}
size_t next_fastq4_record(const char* p, size_t n, size_t pos) {
    if (pos == 0) return 0;
    const char* e = (const char*)memchr(p + pos - 1, '\n', n - (pos - 1));
    if (!e) return n;
    size_t i = (size_t)(e - p) + 1;
    for (int tries = 0; tries < 8 && i < n; ++tries) {            // a line that starts with '@' and whose second successor starts with '+' is a header: a
        const size_t l0 = line_len(p + i, p + n);                  // quality line that starts with '@' is followed by a header and a SEQUENCE line
        const size_t j = i + l0 + 1; if (j >= n) return n;
        const size_t l1 = line_len(p + j, p + n);
        const size_t h = j + l1 + 1;
        if (p[i] == '@' && h < n && p[h] == '+') return i;
        i = j;
    }
    return n;                                                      // (no record start found: the previous slice's parser will say "irregular")
}
void plan_file(const std::string& path, int threads, std::vector<Task>& tasks) {
    if (!is_gzip(path)) {
        auto mp = std::make_shared<Mapped>();
        if (mp->open(path)) {
            const char* p = mp->p; const size_t n = mp->n;
            size_t f0 = 0; while (f0 < n && (p[f0] == '\n' || p[f0] == '\r')) ++f0;
            int kind = 0;
            if (f0 < n && (p[f0] == '>' || p[f0] == ';')) kind = 1;
            else if (f0 < n && p[f0] == '@') {                     // strict four-line FASTQ? (judged by its first record; a slice that finds otherwise reports it)
                const size_t l0 = line_len(p + f0, p + n), s1 = f0 + l0 + 1;
                if (s1 < n) { const size_t l1 = line_len(p + s1, p + n), s2 = s1 + l1 + 1;
                    if (s2 < n && p[s2] == '+') { const size_t l2 = line_len(p + s2, p + n), s3 = s2 + l2 + 1;
                        if (s3 <= n && rstrip_cr(p + s3, s3 < n ? line_len(p + s3, p + n) : 0) == rstrip_cr(p + s1, l1)) kind = 2; } }
            }
            if (kind) {
                size_t slice = std::max<size_t>(n / (size_t)(threads * 4) + 1, 8u << 20);   // a few slices per thread, none below 8 MB
                if (const char* e = getenv("BCALM_SLICE_BYTES")) slice = std::max<size_t>(1, strtoull(e, nullptr, 10));   // (tests: many slices of a small file)
                size_t beg = 0;
                while (beg < n) {
                    size_t end = beg + slice >= n ? n : (kind == 1 ? next_fasta_record(p, n, beg + slice) : next_fastq4_record(p, n, beg + slice));
                    if (end <= beg) end = n;
                    tasks.push_back(Task{ kind, path, mp, beg, end });
                    beg = end;
                }
                return;
            }
        }
    }
    tasks.push_back(Task{ 0, path, nullptr, 0, 0 });
}
// CPUs this process may really use: the affinity mask and the container's CPU quota (cgroup v2 cpu.max, v1 cfs_quota_us / cfs_period_us), not the
// machine's thread count -- a 256-thread host that grants 16 CPUs runs 32 parser threads slower than 16 (profiles/r06_gz_ingest_parallel_inflate.log)
unsigned usable_cpus() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set; CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = std::min<unsigned>(n, (unsigned)c); }
    long long quota = -1, period = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0}; if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else {
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 0; fclose(g); }
    }
    if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
    return n;
}
bool ingest_gz_parallel(Ingest& I, const std::string& path, int threads);
// parse every file with `threads` workers; false: a FASTQ file was not as regular as its first record promised (nothing usable was pushed)
bool ingest_files(Ingest& I, const std::vector<std::string>& files, int threads, bool allow_slices) {
    std::vector<Task> tasks;
    // few gzip files and many threads: each is inflated by all threads, one after the other (pgz.h); many gzip files: one thread each, as before
    std::vector<std::string> gz_parallel;
    size_t n_gz = 0; for (const auto& f : files) n_gz += is_gzip(f);
    const bool pgz_on = allow_slices && threads >= 2 && n_gz * 2 <= (size_t)threads && getenv("BCALM_GZ_SERIAL") == nullptr;
    for (const auto& f : files) {
        if (pgz_on && is_gzip(f)) gz_parallel.push_back(f);
        else if (allow_slices) plan_file(f, threads, tasks);
        else tasks.push_back(Task{ 0, f, nullptr, 0, 0 });
    }
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        try {
            Sink out(I);
            for (;;) {
                const size_t t = next.fetch_add(1);
                if (t >= tasks.size() || I.failed || I.irregular) break;
                const Task& T = tasks[t];
                if (T.kind == 0) parse_stream(T.path, out);
                else if (T.kind == 1) parse_fasta_slice(T.map->p + T.beg, T.map->p + T.end, out);
                else if (!parse_fastq4_slice(T.map->p + T.beg, T.map->p + T.end, out)) { I.irregular = true; break; }
            }
            out.finish();
        } catch (const std::exception& e) { I.fail(e.what()); }
    };
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, tasks.size()));
    std::vector<std::thread> th;
    for (int i = 1; i < nt; ++i) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
    if (I.failed) usage_error(I.error);
    for (const auto& f : gz_parallel) {
        if (I.irregular) break;
        if (!ingest_gz_parallel(I, f, threads)) { Sink out(I); parse_stream(f, out); out.finish(); }
    }
    return !I.irregular;
}

// ---- one gzip file inflated by all threads (pgz.h): real read sets ship as one or two .fastq.gz, and zlib inflates a file with one thread ----
// start of the last record of [p, p + n) that may be incomplete (n: every record is complete); 0: no record start known
size_t last_fasta_start(const char* p, size_t n) {
    for (size_t e = n; e > 1;) {
        const char* q = (const char*)memrchr(p, '>', e); if (!q) return 0;
        const size_t i = (size_t)(q - p);
        if (i == 0 || p[i - 1] == '\n') return i;
        e = i;
    }
    return 0;
}
size_t last_fastq4_start(const char* p, size_t n) {
    for (size_t back = (size_t)1 << 20;; back *= 16) {
        const size_t from = n > back ? n - back : 0;
        size_t r = from ? next_fastq4_record(p, n, from) : 0;
        if (r < n) {
            for (;;) {                                              // whole records from r on: four lines each
                while (r < n && (p[r] == '\n' || p[r] == '\r')) ++r;
                size_t e = r; int lines = 0;
                while (lines < 4) { const char* nl = (const char*)memchr(p + e, '\n', n - e); if (!nl) break; e = (size_t)(nl - p) + 1; ++lines; }
                if (lines < 4) return r;
                r = e;
            }
        }
        if (!from) return 0;
    }
}
struct IrregularInput {};
// false: not handled, nothing was pushed (the caller parses the file with zlib)
bool ingest_gz_parallel(Ingest& I, const std::string& path, int threads) {
    Mapped mp; if (!mp.open(path)) return false;
    // 4 MB of compressed bytes per chunk (its text is held as 16-bit symbols until the wave is resolved); smaller for a small file, so that every thread gets some
    size_t chunk = std::min<size_t>((size_t)4 << 20, std::max<size_t>((size_t)1 << 20, mp.n / ((size_t)threads * 4) + 1));
    if (const char* e = getenv("BCALM_GZ_CHUNK")) chunk = std::max<size_t>(1024, strtoull(e, nullptr, 10));   // (tests: many chunks of a small file)
    std::vector<std::unique_ptr<Sink>> sinks;
    for (int t = 0; t < threads; ++t) sinks.emplace_back(new Sink(I));
    int fmt = 0;                                                    // 1 FASTA, 2 strict four-line FASTQ
    auto on_wave = [&](char* p, size_t n, bool last) -> size_t {
        if (!fmt) {
            size_t f0 = 0; while (f0 < n && (p[f0] == '\n' || p[f0] == '\r')) ++f0;
            if (f0 < n && (p[f0] == '>' || p[f0] == ';')) fmt = 1;
            else if (f0 < n && p[f0] == '@') {                     // (plan_file's test of the first record)
                const size_t l0 = line_len(p + f0, p + n), s1 = f0 + l0 + 1;
                if (s1 < n) { const size_t l1 = line_len(p + s1, p + n), s2 = s1 + l1 + 1;
                    if (s2 < n && p[s2] == '+') { const size_t l2 = line_len(p + s2, p + n), s3 = s2 + l2 + 1;
                        if (s3 <= n && rstrip_cr(p + s3, s3 < n ? line_len(p + s3, p + n) : 0) == rstrip_cr(p + s1, l1)) fmt = 2; } }
            }
            if (!fmt) return pgz::ABORT;
        }
        const size_t cut = last ? n : (fmt == 1 ? last_fasta_start(p, n) : last_fastq4_start(p, n));
        if (!cut) return n;                                         // not one complete record yet: all of it again, with more behind
        std::vector<size_t> b((size_t)threads + 1, cut); b[0] = 0;
        for (int t = 1; t < threads; ++t) {
            const size_t pos = cut / (size_t)threads * (size_t)t;
            const size_t x = pos ? (fmt == 1 ? next_fasta_record(p, cut, pos) : next_fastq4_record(p, cut, pos)) : 0;
            b[t] = std::min(cut, std::max(x, b[t - 1]));
        }
        std::atomic<bool> irregular{false};
        pgz::parallel_for((size_t)threads, threads, [&](size_t t) {
            if (b[t] >= b[t + 1] || I.failed) return;
            try {
                if (fmt == 1) parse_fasta_slice(p + b[t], p + b[t + 1], *sinks[t]);
                else if (!parse_fastq4_slice(p + b[t], p + b[t + 1], *sinks[t])) irregular = true;
            } catch (const std::exception& e) { I.fail(e.what()); }
        });
        if (I.failed) throw std::runtime_error(I.error);
        if (irregular) throw IrregularInput{};
        return n - cut;
    };
    pgz::Stats st; int rc = 0;
    try { rc = pgz::inflate_parallel((const uint8_t*)mp.p, mp.n, threads, chunk, on_wave, &st); }
    catch (const IrregularInput&) { I.irregular = true; return true; }         // (records that are not four lines: the caller starts over)
    catch (const std::exception& e) { usage_error(path + ": " + e.what()); }
    if (rc != 0) return false;
    for (auto& sk : sinks) sk->finish();
    if (getenv("BCALM_GZ_VERBOSE")) fprintf(stderr, "[bcalm] %s: inflated by %d threads, %zu chunks (%zu block starts found), %zu waves, %zu members, %.2f GB of text; find %.2f s, decode %.2f s, resolve %.2f s; parse %.2f s beside the decoding, %.2f s of it waited for\n",
                                            path.c_str(), threads, st.chunks, st.starts_found, st.waves, st.members, st.out_bytes * 1e-9, st.s_find, st.s_decode, st.s_resolve, st.s_caller, st.s_wait);
    return true;
}

// approximate number of sequence bytes a file will deliver (cdbg_expect_input): gzip ~4x, FASTQ carries as many quality bytes
uint64_t estimate_text_bytes(const std::string& path) {
    struct stat st; if (stat(path.c_str(), &st) != 0) return 0;
    uint64_t n = (uint64_t)st.st_size;
    gzFile f = gzopen(path.c_str(), "rb"); char c0 = 0;
    if (f) { const int c = gzgetc(f); if (c >= 0) c0 = (char)c; gzclose(f); }
    FILE* raw = fopen(path.c_str(), "rb"); unsigned char magic[2] = {0, 0};
    if (raw) { if (fread(magic, 1, 2, raw) != 2) magic[0] = 0; fclose(raw); }
    if (magic[0] == 0x1f && magic[1] == 0x8b) n *= 4;
    if (c0 == '@') n /= 2;
    return n + n / 50 + (1 << 20);
}

bool looks_like_file_list(const std::string& path) {       // README.md:47-50 "ls -1 *.fastq > list_reads"
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[4096]; bool is_list = false;
    if (gzgets(f, buf, sizeof buf)) {
        size_t n = strlen(buf); while (n && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) buf[--n] = 0;
        if (n && buf[0] != '>' && buf[0] != '@' && buf[0] != ';') { gzFile g = gzopen(buf, "rb"); if (g) { is_list = true; gzclose(g); } }
    }
    gzclose(f);
    return is_list;
}

void check(int rc) { if (rc != 0) throw std::runtime_error(cdbg_last_error()); }

}  // namespace

int main(int argc, char** argv) {
    if (argc > 1 && (!strcmp(argv[1], "-version") || !strcmp(argv[1], "-v"))) {
        printf("BCALM 2 CLI, MI355X-native engine, version %s\nUsing libcdbg (HIP, gfx950) instead of gatb-core\n", CDBG_VERSION);
        return EXIT_SUCCESS;
    }
    try {
        Options o = parse(argc, argv);
        printf("BCALM 2 CLI, MI355X-native engine, version %s\n", CDBG_VERSION);
        if (o.in.empty()) usage_error("Specifiy -in");             // sic: the reference's message (bcalm_1.cpp:61)
        const std::string prefix = o.out.empty() ? base_name(o.in) : o.out;
        auto t0 = std::chrono::steady_clock::now();

        if (o.n_gpus < 1 || (o.n_gpus & (o.n_gpus - 1))) usage_error("-nb-gpus must be a power of two");
        const int world = o.n_gpus;
        // one context per GPU; with several GPUs the reads are sharded over them and the contexts talk over RCCL inside
        // libcdbg (include/cdbg.h "Multi-GPU"); every rank ends with its SHARE of the unitigs, links the whole set collectively (job-wide ids)
        // and its share is written from its own context -- round 4 gathered the whole graph on every rank (emit_replicated) for rank 0 to write
        std::vector<cdbg_ctx*> ctxs(world, nullptr);
        cdbg_ctx* ctx = nullptr;
        auto make_contexts = [&]() {
            for (int r = 0; r < world; ++r) {
                cdbg_params p{}; p.k = o.k; p.abundance_min = o.amin; p.minimizer_size = o.m; p.log2_partitions = o.log_np;
                p.device_id = world > 1 ? r : o.device; p.world_size = world; p.rank = r; p.all_abundance_counts = o.all_ab ? 1 : 0;
                p.emit_replicated = 0;
                check(cdbg_create(&p, &ctxs[r]));
            }
            ctx = ctxs[0];
            // (CDBG_FORCE_MULTI, the library's test hook: one rank through the multi-rank code path -- it needs the transport then)
            if (world > 1 || getenv("CDBG_FORCE_MULTI")) {
                unsigned char uid[128]; check(cdbg_comm_unique_id(uid));
                std::vector<std::thread> th; std::atomic<int> bad{0}; std::vector<std::string> errs(world);
                for (int r = 0; r < world; ++r) th.emplace_back([&, r]() { if (cdbg_comm_init_rccl(ctxs[r], uid) != 0) { errs[r] = cdbg_last_error(); ++bad; } });
                for (auto& t : th) t.join();
                if (bad) for (auto& e : errs) if (!e.empty()) usage_error(e);
            }
        };
        make_contexts();
        auto t_init = std::chrono::steady_clock::now();
        int threads = o.cores > 0 ? o.cores : (int)std::min<unsigned>(usable_cpus(), 32u);
        threads = std::max(1, std::min(threads, 60));        // (each parser thread holds one of the library's 64 staging buffers)
        std::vector<std::string> files;
        if (looks_like_file_list(o.in)) {
            gzFile f = gzopen(o.in.c_str(), "rb"); char buf[4096];
            while (gzgets(f, buf, sizeof buf)) { size_t n = strlen(buf); while (n && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) buf[--n] = 0; if (n) files.push_back(buf); }
            gzclose(f);
        } else files.push_back(o.in);
        uint64_t n_seq = 0, n_bases = 0;
        for (int attempt = 0;; ++attempt) {
            if (world == 1 && !o.no_stream) {                // announce the volume: the scan starts while the input is still being parsed
                uint64_t est = 0; for (const auto& f : files) est += estimate_text_bytes(f);
                if (est) check(cdbg_expect_input(ctx, est));
            }
            Ingest I; I.ctxs = ctxs; I.k = o.k;
            const bool regular = ingest_files(I, files, threads, attempt == 0);
            n_seq = I.n_seq; n_bases = I.n_bases;
            if (regular) break;
            // a FASTQ file whose records are not all four lines: start over with the tolerant serial parser
            for (int r = 0; r < world; ++r) { cdbg_destroy(ctxs[r]); ctxs[r] = nullptr; }
            make_contexts();
        }
        auto t1 = std::chrono::steady_clock::now();
        // the stages are collective: one host thread per GPU
        auto all_ranks = [&](int (*stage)(cdbg_ctx*)) {
            if (world == 1) { check(stage(ctx)); return; }
            std::vector<std::thread> th; std::atomic<int> bad{0}; std::vector<std::string> errs(world);
            for (int r = 0; r < world; ++r) th.emplace_back([&, r]() { if (stage(ctxs[r]) != 0) { errs[r] = cdbg_last_error(); ++bad; } });
            for (auto& t : th) t.join();
            if (bad) for (auto& e : errs) if (!e.empty()) throw std::runtime_error(e);
        };
        all_ranks(cdbg_count);
        if (!o.solid_out.empty()) {                          // solid k-mer dump: "<canonical k-mer> <abundance>" per line
            FILE* sf = fopen(o.solid_out.c_str(), "w");
            if (!sf) usage_error("cannot write " + o.solid_out);
            for (int r = 0; r < world; ++r) {                // (every rank holds the solid k-mers of its partitions)
                uint64_t ns = 0, got = 0; check(cdbg_num_solid(ctxs[r], &ns));
                std::vector<char> km((ns + 1) * (size_t)(o.k + 1)); std::vector<uint32_t> cnt(ns + 1);
                check(cdbg_fetch_solid(ctxs[r], km.data(), cnt.data(), ns, &got));
                for (uint64_t i = 0; i < got; ++i) fprintf(sf, "%s %u\n", km.data() + i * (size_t)(o.k + 1), cnt[i]);
            }
            fclose(sf);
        }
        all_ranks(cdbg_compact);
        all_ranks(cdbg_glue);
        auto t_stages = std::chrono::steady_clock::now();
        cdbg_stats_t st; check(cdbg_stats(ctx, &st));
        // edges between unitigs (README.md:72 L: tokens; convertToGFA.py:103-112 GFA L lines).  Several GPUs: every rank holds a share of
        // the unitigs; cdbg_link is collective there and numbers the unitigs job-wide, rank after rank (include/cdbg.h)
        all_ranks(cdbg_link);
        for (int r = 1; r < world; ++r) {                    // (the printed totals: k-mers and pieces of every rank's partitions, unitigs of every rank's share)
            cdbg_stats_t sr; check(cdbg_stats(ctxs[r], &sr));
            st.n_occurrences += sr.n_occurrences; st.n_distinct += sr.n_distinct; st.n_solid += sr.n_solid; st.n_pieces += sr.n_pieces;
            st.n_unitigs += sr.n_unitigs; st.unitig_bases += sr.unitig_bases;
            st.ms_count = std::max(st.ms_count, sr.ms_count); st.ms_compact = std::max(st.ms_compact, sr.ms_compact); st.ms_glue = std::max(st.ms_glue, sr.ms_glue);
        }

        // Output: rank after rank (one rank: everything), each rank's share fetched, formatted by the threads in blocks of unitigs (own
        // integer formatting; "%.1f" stays with printf so that km:f: rounds exactly as the reference's) and written in block order, one
        // write per block.  Ids and link targets are job-wide.
        const std::string fa = prefix + ".unitigs.fa";
        FILE* out = fopen(fa.c_str(), "w");
        if (!out) usage_error("cannot write " + fa);
        FILE* gfa = nullptr;
        if (o.gfa) { gfa = fopen((prefix + ".unitigs.gfa").c_str(), "w"); if (gfa) fprintf(gfa, "H\tVN:Z:1.0\tks:i:%d\n", o.k); }
        double s_fetch = 0, s_write = 0;
        for (int r = 0; r < world; ++r) {
        auto tf0 = std::chrono::steady_clock::now();
        cdbg_ctx* const cx = ctxs[r];
        uint64_t nu = 0, tb = 0; check(cdbg_num_unitigs(cx, &nu, &tb));
        uint64_t id0 = 0, id_total = 0; check(cdbg_unitig_id_base(cx, &id0, &id_total));
        std::unique_ptr<char[]> seq(new char[tb + 1]);       // (not value-initialised: a gigabyte at config 3)
        std::vector<uint64_t> off(nu + 1), kc(nu ? nu : 1);
        check(cdbg_fetch_unitigs(cx, 0, nu, seq.get(), off.data(), kc.data()));
        std::vector<uint32_t> ab; std::vector<uint64_t> aboff;
        if (o.all_ab) { ab.resize(tb + 1); aboff.resize(nu + 1); check(cdbg_fetch_unitig_abundances(cx, 0, nu, ab.data(), aboff.data())); }
        uint64_t nl = 0; check(cdbg_num_links(cx, &nl));
        std::vector<uint64_t> loff(2 * nu + 1); std::vector<uint32_t> lto(nl ? nl : 1);
        check(cdbg_fetch_links(cx, loff.data(), lto.data()));
        auto tf1 = std::chrono::steady_clock::now();
        {
            const uint64_t BLOCK = 1u << 15;
            const uint64_t nblocks = (nu + BLOCK - 1) / BLOCK;
            std::atomic<uint64_t> next_block{0};
            std::mutex wm; std::condition_variable wcv; uint64_t turn = 0; bool wfail = false;
            auto put_u = [](std::string& d, unsigned long long v) { char t[24]; int n = 0; do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v); while (n) d.push_back(t[--n]); };
            auto put_km = [](std::string& d, double km) { char t[48]; const int n = snprintf(t, sizeof t, "%.1f", km); d.append(t, (size_t)n); };
            auto writer = [&]() {
                std::string fb, gb;
                for (;;) {
                    const uint64_t b = next_block.fetch_add(1);
                    if (b >= nblocks) break;
                    fb.clear(); gb.clear();
                    const uint64_t i0 = b * BLOCK, i1 = std::min(nu, i0 + BLOCK);
                    fb.reserve((size_t)(off[i1] - off[i0]) + (size_t)(i1 - i0) * 96);
                    for (uint64_t i = i0; i < i1; ++i) {
                        const uint64_t len = off[i + 1] - off[i], id = id0 + i;
                        const double km = (double)kc[i] / (double)(len - (uint64_t)o.k + 1);
                        fb.push_back('>'); put_u(fb, id); fb.append(" LN:i:"); put_u(fb, len);
                        if (o.all_ab) {                          // ><id> LN:i:<length> ab:Z:<abundance_0> ... (README.md:76)
                            fb.append(" ab:Z:");
                            for (uint64_t j = aboff[i]; j < aboff[i + 1]; ++j) { if (j != aboff[i]) fb.push_back(' '); put_u(fb, ab[j]); }
                        } else { fb.append(" KC:i:"); put_u(fb, kc[i]); fb.append(" km:f:"); put_km(fb, km); }
                        if (gfa) { gb.append("S\t"); put_u(gb, id); gb.push_back('\t'); gb.append(seq.get() + off[i], (size_t)len);
                                   gb.append("\tLN:i:"); put_u(gb, len); gb.append("\tKC:i:"); put_u(gb, kc[i]); gb.append("\tkm:f:"); put_km(gb, km); gb.push_back('\n'); }
                        for (int side = 1; side >= 0; --side)    // '+' links (through the last k-mer) first, then '-'
                            for (uint64_t j = loff[2 * i + side]; j < loff[2 * i + side + 1]; ++j) {
                                const char fs = side ? '+' : '-', ts = (lto[j] & 1u) ? '-' : '+';
                                fb.append(" L:"); fb.push_back(fs); fb.push_back(':'); put_u(fb, lto[j] >> 1); fb.push_back(':'); fb.push_back(ts);
                                if (gfa) { gb.append("L\t"); put_u(gb, id); gb.push_back('\t'); gb.push_back(fs); gb.push_back('\t'); put_u(gb, lto[j] >> 1);
                                           gb.push_back('\t'); gb.push_back(ts); gb.push_back('\t'); put_u(gb, (unsigned long long)(o.k - 1)); gb.append("M\n"); }
                            }
                        fb.append(" \n");
                        fb.append(seq.get() + off[i], (size_t)len); fb.push_back('\n');
                    }
                    std::unique_lock<std::mutex> l(wm);
                    wcv.wait(l, [&] { return turn == b; });
                    if (fwrite(fb.data(), 1, fb.size(), out) != fb.size()) wfail = true;
                    if (gfa && fwrite(gb.data(), 1, gb.size(), gfa) != gb.size()) wfail = true;
                    ++turn; wcv.notify_all();
                }
            };
            const int nt = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)threads, nblocks));
            std::vector<std::thread> th;
            for (int i = 1; i < nt; ++i) th.emplace_back(writer);
            writer();
            for (auto& t : th) t.join();
            if (wfail) usage_error("write error on " + fa);
        }
        auto tf2 = std::chrono::steady_clock::now();
        s_fetch += std::chrono::duration<double>(tf1 - tf0).count(); s_write += std::chrono::duration<double>(tf2 - tf1).count();
        }
        fclose(out); if (gfa) fclose(gfa);
        auto t3 = std::chrono::steady_clock::now();
        const auto t2 = t3 - std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(s_write));   // (fetches and writes alternate with several ranks: the printed split is their sums)
        auto sec = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
        printf("input: %llu sequences, %llu bases (%.2f s parse)\n", (unsigned long long)n_seq, (unsigned long long)n_bases, sec(t_init, t1));
        uint64_t in_bytes = 0; for (const auto& f : files) { struct stat sb; if (stat(f.c_str(), &sb) == 0) in_bytes += (uint64_t)sb.st_size; }
        uint64_t out_bytes = 0; { struct stat sb; if (stat(fa.c_str(), &sb) == 0) out_bytes = (uint64_t)sb.st_size; }
        printf("host: init %.2f s; ingest %.2f s = %.2f GB/s of input files (%d threads: parse into pinned buffers + H2D + the overlapped scan); stages %.2f s; links + D2H %.2f s; write %.2f s = %.2f GB/s (%d threads)\n",
               sec(t0, t_init), sec(t_init, t1), (double)in_bytes / 1e9 / std::max(1e-9, sec(t_init, t1)), threads, sec(t1, t_stages), sec(t_stages, t2), sec(t2, t3),
               (double)out_bytes / 1e9 / std::max(1e-9, sec(t2, t3)), threads);
        printf("k-mers: %llu occurrences, %llu distinct, %llu solid (abundance >= %d)\n", (unsigned long long)st.n_occurrences,
               (unsigned long long)st.n_distinct, (unsigned long long)st.n_solid, o.amin);
        printf("graph: %llu pieces -> %llu unitigs, %llu bases; minimizer size %d, 2^%d partitions\n", (unsigned long long)st.n_pieces,
               (unsigned long long)st.n_unitigs, (unsigned long long)st.unitig_bases, st.minimizer_size, st.log2_partitions);
        printf("GPU: scan %.2f+%.2f ms (%llu of %llu tiles scanned while the input was arriving), count %.2f ms, compact %.2f ms, glue %.2f ms; stages+links+D2H %.2f s; write %.2f s; end to end %.2f s\n",
               st.ms_scan_hist, st.ms_scan_emit, (unsigned long long)st.n_tiles_overlapped, (unsigned long long)st.n_launch_scan, st.ms_count, st.ms_compact, st.ms_glue, sec(t1, t2), sec(t2, t3), sec(t0, t3));
        printf("unitigs written to %s\n", fa.c_str());
        for (cdbg_ctx* x : ctxs) cdbg_destroy(x);
    } catch (const std::exception& e) {
        printf("EXCEPTION: %s\n", e.what());
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
