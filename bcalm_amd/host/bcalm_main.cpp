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

#include <sys/stat.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "cdbg.h"

#ifndef CDBG_VERSION
#define CDBG_VERSION "cdbg-mi355x r1 (CLI-compatible with BCALM 2 v2.2.3)"
#endif

namespace {

struct Options {
    std::string in, out;
    int k = 31, amin = 2, m = 0, device = 0, log_np = -1, n_gpus = 1;
    bool gfa = false, verbose = false, all_ab = false;
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
        else if (a == "-all-abundance-counts") o.all_ab = true;        // README.md:74-80
        else if (a == "-solid-kmers-out") o.solid_out = need("-solid-kmers-out");   // hidden in the reference (bcalm_1.cpp:37)
        else if (a == "-verbose") { o.verbose = true; if (i + 1 < argc && argv[i + 1][0] != '-') ++i; }
        else if (a == "-nb-cores" || a == "-max-memory" || a == "-max-disk" || a == "-out-tmp" || a == "-out-dir" ||
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

// Two-stage ingest: a parser thread turns FASTA / FASTQ (plain or gzip: zlib reads both) into chunks of sequences
// separated by '\n'; the main thread pushes the chunks (cdbg_push_text copies into pinned staging buffers, the H2D
// copies and -- with cdbg_expect_input -- the read scan itself overlap the parsing of the next chunk).
struct ChunkQueue {
    std::mutex m; std::condition_variable cv; std::deque<std::string> q; bool done = false; std::string error;
    void put(std::string&& c) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return q.size() < 6; }); q.push_back(std::move(c)); cv.notify_all(); }
    bool get(std::string& c) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !q.empty() || done; }); if (q.empty()) return false; c = std::move(q.front()); q.pop_front(); cv.notify_all(); return true; }
    void finish(const std::string& err = "") { std::lock_guard<std::mutex> l(m); done = true; if (!err.empty()) error = err; cv.notify_all(); }
};
void parse_file(const std::string& path, ChunkQueue& out, size_t chunk_bytes, uint64_t& n_seq, uint64_t& n_bases) {
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) usage_error("cannot open input file " + path);
    gzbuffer(f, 1 << 20);
    std::vector<char> line(1 << 22);
    std::string chunk; chunk.reserve(chunk_bytes + (1 << 20));
    int fmt = 0;                                // 0 unknown, 1 FASTA, 2 FASTQ
    // FASTQ records may wrap their sequence and quality over several lines (README.md:45-50 accepts any FASTQ): after the
    // '@' header the sequence runs until the '+' line, the quality until it is as long as the sequence
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
                    else { gzclose(f); usage_error("malformed FASTQ record in " + path + " (sequence " + std::to_string(n_seq + 1) + "): '@' expected"); }
                } else if (fq_state == 1) fq_kind = (n && line[0] == '+') ? 'P' : 'S';
                else fq_kind = 'Q';
            }
            if (fq_kind == 'S') { chunk.append(line.data(), n); n_bases += n; fq_seq += n; }
            else if (fq_kind == 'Q') fq_qual += n;
            if (complete) {
                if (fq_kind == 'H') { fq_state = 1; fq_seq = 0; }
                else if (fq_kind == 'P') { chunk.push_back('\n'); ++n_seq; fq_state = 2; fq_qual = 0; }
                else if (fq_kind == 'Q' && fq_qual >= fq_seq) fq_state = 0;
            }
        } else {
            if (starts_line && n && line[0] == '>') { if (!chunk.empty() && chunk.back() != '\n') chunk.push_back('\n'); ++n_seq; }
            else if (!(starts_line && n && line[0] == ';')) { chunk.append(line.data(), n); n_bases += n; }
        }
        if (chunk.size() > chunk_bytes && chunk.back() == '\n') { out.put(std::move(chunk)); chunk.clear(); chunk.reserve(chunk_bytes + (1 << 20)); }
    }
    gzclose(f);
    if (!chunk.empty() && chunk.back() != '\n') chunk.push_back('\n');
    if (!chunk.empty()) out.put(std::move(chunk));
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
        // libcdbg (include/cdbg.h "Multi-GPU"); rank 0 emits the complete unitig set (emit_replicated) and writes the file
        std::vector<cdbg_ctx*> ctxs(world, nullptr);
        for (int r = 0; r < world; ++r) {
            cdbg_params p{}; p.k = o.k; p.abundance_min = o.amin; p.minimizer_size = o.m; p.log2_partitions = o.log_np;
            p.device_id = world > 1 ? r : o.device; p.world_size = world; p.rank = r; p.all_abundance_counts = o.all_ab ? 1 : 0;
            p.emit_replicated = 1;
            check(cdbg_create(&p, &ctxs[r]));
        }
        cdbg_ctx* ctx = ctxs[0];
        if (world > 1) {
            unsigned char uid[128]; check(cdbg_comm_unique_id(uid));
            std::vector<std::thread> th; std::atomic<int> bad{0}; std::vector<std::string> errs(world);
            for (int r = 0; r < world; ++r) th.emplace_back([&, r]() { if (cdbg_comm_init_rccl(ctxs[r], uid) != 0) { errs[r] = cdbg_last_error(); ++bad; } });
            for (auto& t : th) t.join();
            if (bad) for (auto& e : errs) if (!e.empty()) usage_error(e);
        }
        uint64_t n_seq = 0, n_bases = 0; size_t next_ctx = 0;
        std::vector<std::string> files;
        if (looks_like_file_list(o.in)) {
            gzFile f = gzopen(o.in.c_str(), "rb"); char buf[4096];
            while (gzgets(f, buf, sizeof buf)) { size_t n = strlen(buf); while (n && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) buf[--n] = 0; if (n) files.push_back(buf); }
            gzclose(f);
        } else files.push_back(o.in);
        if (world == 1) {                                    // announce the volume: the scan starts while the input is still being parsed
            uint64_t est = 0; for (const auto& f : files) est += estimate_text_bytes(f);
            if (est) check(cdbg_expect_input(ctx, est));
        }
        {
            ChunkQueue q;
            std::thread parser([&]() {
                try { for (const auto& f : files) parse_file(f, q, world > 1 ? (4u << 20) : (32u << 20), n_seq, n_bases); q.finish(); }
                catch (const std::exception& e) { q.finish(e.what()); }
            });
            std::string chunk; std::string push_err;
            while (q.get(chunk)) {                           // (several GPUs: whole chunks of complete sequences go to the contexts in turn)
                if (push_err.empty() && cdbg_push_text(ctxs[next_ctx], chunk.data(), chunk.size()) != 0) push_err = cdbg_last_error();
                next_ctx = (next_ctx + 1) % ctxs.size();
            }
            parser.join();
            if (!q.error.empty()) usage_error(q.error);
            if (!push_err.empty()) usage_error(push_err);
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
        cdbg_stats_t st; check(cdbg_stats(ctx, &st));
        uint64_t nu = 0, tb = 0; check(cdbg_num_unitigs(ctx, &nu, &tb));
        std::vector<char> seq(tb + 1); std::vector<uint64_t> off(nu + 1), kc(nu ? nu : 1);
        check(cdbg_fetch_unitigs(ctx, 0, nu, seq.data(), off.data(), kc.data()));
        std::vector<uint32_t> ab; std::vector<uint64_t> aboff;
        if (o.all_ab) { ab.resize(tb + 1); aboff.resize(nu + 1); check(cdbg_fetch_unitig_abundances(ctx, 0, nu, ab.data(), aboff.data())); }
        // edges between unitigs (README.md:72 L: tokens; convertToGFA.py:103-112 GFA L lines)
        check(cdbg_link(ctx));
        uint64_t nl = 0; check(cdbg_num_links(ctx, &nl));
        std::vector<uint64_t> loff(2 * nu + 1); std::vector<uint32_t> lto(nl ? nl : 1);
        check(cdbg_fetch_links(ctx, loff.data(), lto.data()));
        auto t2 = std::chrono::steady_clock::now();

        const std::string fa = prefix + ".unitigs.fa";
        FILE* out = fopen(fa.c_str(), "w");
        if (!out) usage_error("cannot write " + fa);
        FILE* gfa = nullptr;
        if (o.gfa) { gfa = fopen((prefix + ".unitigs.gfa").c_str(), "w"); if (gfa) fprintf(gfa, "H\tVN:Z:1.0\tks:i:%d\n", o.k); }
        for (uint64_t i = 0; i < nu; ++i) {
            const uint64_t len = off[i + 1] - off[i];
            const double km = (double)kc[i] / (double)(len - (uint64_t)o.k + 1);
            if (o.all_ab) {                                  // ><id> LN:i:<length> ab:Z:<abundance_0> ... (README.md:76)
                fprintf(out, ">%llu LN:i:%llu ab:Z:", (unsigned long long)i, (unsigned long long)len);
                for (uint64_t j = aboff[i]; j < aboff[i + 1]; ++j) fprintf(out, j == aboff[i] ? "%u" : " %u", ab[j]);
            } else
            fprintf(out, ">%llu LN:i:%llu KC:i:%llu km:f:%.1f", (unsigned long long)i, (unsigned long long)len, (unsigned long long)kc[i], km);
            if (gfa) { fprintf(gfa, "S\t%llu\t", (unsigned long long)i); fwrite(seq.data() + off[i], 1, len, gfa);
                       fprintf(gfa, "\tLN:i:%llu\tKC:i:%llu\tkm:f:%.1f\n", (unsigned long long)len, (unsigned long long)kc[i], km); }
            for (int side = 1; side >= 0; --side)            // '+' links (through the last k-mer) first, then '-'
                for (uint64_t j = loff[2 * i + side]; j < loff[2 * i + side + 1]; ++j) {
                    const char fs = side ? '+' : '-', ts = (lto[j] & 1u) ? '-' : '+';
                    fprintf(out, " L:%c:%u:%c", fs, lto[j] >> 1, ts);
                    if (gfa) fprintf(gfa, "L\t%llu\t%c\t%u\t%c\t%dM\n", (unsigned long long)i, fs, lto[j] >> 1, ts, o.k - 1);
                }
            fputs(" \n", out);
            fwrite(seq.data() + off[i], 1, len, out); fputc('\n', out);
        }
        fclose(out); if (gfa) fclose(gfa);
        auto t3 = std::chrono::steady_clock::now();
        auto sec = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
        printf("input: %llu sequences, %llu bases (%.2f s parse)\n", (unsigned long long)n_seq, (unsigned long long)n_bases, sec(t0, t1));
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
