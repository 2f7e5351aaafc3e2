// pgz_cat -- the parallel gzip inflater of pgz.h as a filter (test and measurement tool; pure host code):
//   pgz_cat <file.gz> [threads] [chunk bytes]   -> the inflated bytes on stdout, one line of statistics on stderr
// exit status 0: done; 2: "not handled" (pgz.h: the caller would inflate with zlib); 1: error.
#include "pgz.h"
#include <chrono>
#include <cstdio>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: pgz_cat <file.gz> [threads] [chunk bytes]\n"); return 1; }
    const int threads = argc > 2 ? atoi(argv[2]) : (int)std::thread::hardware_concurrency();
    const size_t chunk = argc > 3 ? strtoull(argv[3], nullptr, 10) : (size_t)4 << 20;
    const bool quiet = getenv("PGZ_NO_OUTPUT") != nullptr;
    const int fd = open(argv[1], O_RDONLY); struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0 || st.st_size == 0) { fprintf(stderr, "pgz_cat: cannot open %s\n", argv[1]); return 1; }
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { fprintf(stderr, "pgz_cat: cannot map %s\n", argv[1]); return 1; }
    const auto t0 = std::chrono::steady_clock::now();
    pgz::Stats s; int rc;
    try {
        rc = pgz::inflate_parallel((const uint8_t*)m, (size_t)st.st_size, threads, chunk, [&](char* p, size_t n, bool) -> size_t {
            if (!quiet && fwrite(p, 1, n, stdout) != n) throw std::runtime_error("write failed");
            return 0;
        }, &s);
    } catch (const std::exception& e) { fprintf(stderr, "pgz_cat: %s\n", e.what()); return 1; }
    if (rc != 0) { fprintf(stderr, "pgz_cat: not handled\n"); return 2; }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "{\"threads\": %d, \"chunks\": %zu, \"starts_found\": %zu, \"chunks_on_chain\": %zu, \"waves\": %zu, \"members\": %zu, \"in_bytes\": %zu, \"out_bytes\": %llu, \"seconds\": %.3f, \"out_GB_per_s\": %.3f, \"s_find\": %.3f, \"s_decode\": %.3f, \"s_resolve\": %.3f, \"s_caller\": %.3f, \"s_wait_for_caller\": %.3f}\n",
            threads, s.chunks, s.starts_found, s.chunks_on_chain, s.waves, s.members, (size_t)st.st_size, (unsigned long long)s.out_bytes, sec, s.out_bytes / sec * 1e-9, s.s_find, s.s_decode, s.s_resolve, s.s_caller, s.s_wait);
    return 0;
}
