// bcalm_tools -- the reference's downstream helpers as one native host program
// (post-processing parity; pure host code, no GPU involved).
//
//   bcalm_tools convertToGFA   <unitigs.fa> <out.gfa> <k> [-s|--single-directed]
//   bcalm_tools split_unitigs  <references.fa> <unitigs.fa> <k>      -> <unitigs.fa>.split.fa
//   bcalm_tools pufferize      <references.fa> <unitigs.fa> <k>      -> <unitigs.fa>.pufferized.gfa
//   bcalm_tools abundance_stats <unitigs.fa>
//
// Behaviour restated from (files written, their bytes, exit status and the messages on stdout):
//   /root/reference/scripts/convertToGFA.py:35-120     (H line :68, S line :35-47, L lines :103-112, -s :105-109)
//   /root/reference/scripts/split_unitigs.py:51-110    (cut rules :92-103, renumbering :67-71, warnings :80-87)
//   /root/reference/scripts/pufferize.py:50-137        (S lines :65-69, abort on a repeated end k-mer :78-81,
//                                                       P lines :104-133; only the END k-mer map is ever filled,
//                                                       :82, so every path step is reported with '-' and a k-mer that
//                                                       is not a unitig's last k-mer aborts the run, :128-130)
//   /root/reference/scripts/abundance_stats.py:26-41   (km:f: truncated to int, count and total length per value)
// tests/test_postproc.py pins every command byte-for-byte against outputs of those scripts
// (tests/golden/postproc/, made by tests/golden/make_postproc_golden.py).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

[[noreturn]] void die(const std::string& msg) {          // python: exit("text") -> text on stderr, status 1
    fflush(stdout);
    fprintf(stderr, "%s\n", msg.c_str());
    exit(1);
}

const char* WS = " \t\n\r\f\v";
std::string strip(const std::string& s) {
    const size_t a = s.find_first_not_of(WS);
    if (a == std::string::npos) return "";
    return s.substr(a, s.find_last_not_of(WS) - a + 1);
}
std::string rstrip(const std::string& s) {
    const size_t b = s.find_last_not_of(WS);
    return b == std::string::npos ? "" : s.substr(0, b + 1);
}

struct Record { std::string header, seq; };

// header line without '>' and stripped, sequence lines stripped and joined; of several header lines
// in a row the first one names the record (what the scripts' groupby-based reader does)
std::vector<Record> read_fasta(const std::string& path) {
    std::ifstream f(path);
    if (!f) die("cannot open " + path);
    std::vector<Record> out;
    std::string line; bool in_seq = false, have = false;
    while (std::getline(f, line)) {
        if (!line.empty() && line[0] == '>') {
            if (!have || in_seq) { out.push_back({ strip(line.substr(1)), "" }); have = true; in_seq = false; }
        } else if (have) {
            out.back().seq += strip(line); in_seq = true;
        }
    }
    if (have && !in_seq) out.pop_back();                    // header without a sequence group
    return out;
}

char comp(char c) { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return c; } }
std::string revcomp(const std::string& s) { std::string r(s.rbegin(), s.rend()); for (char& c : r) c = comp(c); return r; }
std::string normalize(const std::string& s) { std::string r = revcomp(s); return s < r ? s : r; }
std::string head(const std::string& s, size_t k) { return s.substr(0, std::min(k, s.size())); }
std::string tail(const std::string& s, size_t k) { return s.size() > k ? s.substr(s.size() - k) : s; }

// ---------------------------------------------------------------- convertToGFA
int to_gfa(int argc, char** argv) {
    std::vector<std::string> pos; bool single = false;
    for (int i = 0; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-s" || a == "--single-directed") single = true; else pos.push_back(a);
    }
    if (pos.size() != 3) die("usage: bcalm_tools convertToGFA [-s] inputFilename outputFilename kmerSize");
    const int k = atoi(pos[2].c_str());
    std::ifstream f(pos[0]);
    if (!f) die("cannot open " + pos[0]);
    FILE* g = fopen(pos[1].c_str(), "w");
    if (!g) die("cannot write " + pos[1]);
    fprintf(g, "H\tVN:Z:1.0\tks:i:%d\n", k);
    printf("GFA file open\n");
    std::string name, segment; std::vector<std::string> optional, links;
    auto flush_segment = [&]() {
        std::string add = "S\t" + name + "\t" + segment + "\t";
        for (const std::string& o : optional) add += o + "\t";
        add = rstrip(add);
        fprintf(g, "%s\n", add.c_str());
        for (const std::string& l : links) fputs(l.c_str(), g);
    };
    bool first = true; std::string line;
    const std::string ov = std::to_string(k - 1) + "M\n";
    while (std::getline(f, line)) {
        if (line.empty()) continue;
        if (line[0] != '>') { segment += line; continue; }
        if (!first) { flush_segment(); segment.clear(); }
        first = false; optional.clear(); links.clear();
        std::vector<std::string> a; size_t p = 0;
        for (;;) { const size_t q = line.find(' ', p); a.push_back(line.substr(p, q == std::string::npos ? q : q - p)); if (q == std::string::npos) break; p = q + 1; }
        name = a[0].substr(1);
        for (size_t i = 1; i < a.size(); ++i) {
            const std::string& t = a[i];
            if (t.empty()) continue;
            if (t.compare(0, 2, "MA") == 0) optional.push_back("MA:f:" + t.substr(2));
            else if (t.compare(0, 2, "L:") == 0) {
                std::vector<std::string> b; size_t s = 0;
                for (;;) { const size_t q = t.find(':', s); b.push_back(t.substr(s, q == std::string::npos ? q : q - s)); if (q == std::string::npos) break; s = q + 1; }
                if (b.size() < 4) die("malformed link token " + t);
                const std::string l = "L\t" + name + "\t" + b[1] + "\t" + b[2] + "\t" + b[3] + "\t" + ov;
                if (!single) links.push_back(l);
                else if (name < b[2]) links.push_back(l);                    // string order, as the script compares ids
                else if (name == b[2] && !(b[1] == "-" && b[3] == "-")) links.push_back(l);
            } else optional.push_back(t);
        }
    }
    flush_segment();
    printf("done\n");
    fclose(g);
    return 0;
}

// ---------------------------------------------------------------- split_unitigs / pufferize
struct EndInfo { long id; size_t len; };
struct Splitter {
    bool puffer; size_t k; FILE* out; long n = -1;
    std::unordered_set<std::string> ref_s, ref_e;
    std::unordered_map<std::string, EndInfo> u_s, u_e;

    static std::string show(const EndInfo& e) { return "[" + std::to_string(e.id) + ", " + std::to_string(e.len) + "]"; }

    void create(std::string u) {
        if (u.size() == k) u = normalize(u);
        ++n;
        if (puffer) fprintf(out, "S\t%ld\t%s\n", n, u.c_str()); else fprintf(out, ">unitig%ld\n%s\n", n, u.c_str());
        const std::string hs = head(u, k), ts = tail(u, k), h = normalize(hs), t = normalize(ts);
        if (puffer) {
            if (u_s.count(h) || u_e.count(h)) { fflush(out); die("Error: Initial kmer is repeated."); }
            if (u_s.count(t) || u_e.count(t)) { fflush(out); die("Error: Last kmer is repeated."); }
            u_e[t] = { n, u.size() };
        } else {
            if (u_s.count(h)) printf("Warning, start kmer (%s) was also seen at start of unitig %s.\n", hs.c_str(), show(u_s[h]).c_str());
            if (u_e.count(h)) printf("Warning, start kmer (%s) was also seen at end of unitig %s.\n", hs.c_str(), show(u_e[h]).c_str());
            if (u_s.count(t)) printf("Warning, last kmer (%s) was also seen at start of unitig %s.\n", ts.c_str(), show(u_s[t]).c_str());
            if (u_e.count(t)) printf("Warning, last kmer (%s) was also seen at end of unitig %s.\n", ts.c_str(), show(u_e[t]).c_str());
            u_e[t] = { n, u.size() };
            u_s[h] = { n, u.size() };
        }
    }
    void split(const std::string& u) {
        size_t prev = 0;
        if (u.size() >= k)
            for (size_t i = 0; i + k <= u.size(); ++i) {
                const std::string kmer = u.substr(i, k), rc = revcomp(kmer);
                if (ref_s.count(kmer) || ref_e.count(rc))                       // a reference starts here: cut before the k-mer
                    if (i + k - 1 >= prev + k) { create(u.substr(prev, i + k - 1 - prev)); prev = i; }
                if (ref_e.count(kmer) || ref_s.count(rc)) {                     // a reference ends here: cut after the k-mer
                    create(u.substr(prev, i + k - prev)); prev = i + 1;
                }
            }
        if (u.size() >= prev + k) create(u.substr(prev));
    }
};

int split_main(bool puffer, int argc, char** argv) {
    if (argc < 3) {
        if (puffer) printf("alters BCALM's unitigs so that they fit pufferfish input: unitigs are split at k-mers that are extremities of the reference sequences\n");
        else printf("split BCALM unitigs at reference extremities: each first k-mer of a reference starts a unitig, each last k-mer ends one\n");
        die("arguments: references.fa unitigs.fa k");
    }
    const std::string references = argv[0], unitigs = argv[1];
    Splitter s; s.puffer = puffer; s.k = (size_t)atoi(argv[2]);
    const std::vector<Record> refs = read_fasta(references);
    for (const Record& r : refs) { s.ref_s.insert(head(r.seq, s.k)); s.ref_e.insert(tail(r.seq, s.k)); }
    const std::string outname = unitigs + (puffer ? ".pufferized.gfa" : ".split.fa");
    s.out = fopen(outname.c_str(), "w");
    if (!s.out) die("cannot write " + outname);
    printf("Start parsing and spliting unitigs .. \n");
    for (const Record& u : read_fasta(unitigs)) s.split(u.seq);
    if (!puffer) {
        fclose(s.out);
        printf("done. result is in: %s\n", outname.c_str());
        return 0;
    }
    printf("Start reconstructing the path .. \n");
    long paths = 0;
    for (const Record& r : refs) {
        fputs("\nP\t", s.out);
        const std::string& ref = r.seq;
        size_t i = 0;
        while (i + s.k <= ref.size()) {
            const std::string kmer = ref.substr(i, s.k), nk = normalize(kmer);
            const EndInfo* e = nullptr; char ori = '+';
            const auto is = s.u_s.find(nk), ie = s.u_e.find(nk);
            if (is != s.u_s.end() && ie != s.u_e.end()) { e = &is->second; ori = kmer == nk ? '+' : '-'; }
            else if (is != s.u_s.end()) { e = &is->second; ori = '+'; }
            else if (ie != s.u_e.end()) { e = &ie->second; ori = '-'; }
            else {
                printf("%ld  paths reconstructed.\n", paths);
                fflush(s.out);
                die("ERROR: kmer is not found in the start or end of a unitig \n" + kmer + "  ,  " + nk);
            }
            fprintf(s.out, "%ld%c,", e->id, ori);
            i += e->len - s.k + 1;                                             // skip the k-mers of that unitig
        }
        ++paths;
    }
    fclose(s.out);
    printf("done. result is in: %s\n", outname.c_str());
    printf("to get a GFA file with links for the split unitigs, re-run the link step on them and then:\n");
    printf("bcalm_tools convertToGFA %s %s.gfa %zu\n", unitigs.c_str(), unitigs.c_str(), s.k);
    return 0;
}

// ---------------------------------------------------------------- abundance_stats
int abundance_main(int argc, char** argv) {
    if (argc < 1) {
        printf("prints some abundance statitics of a unitigs FASTA file produced by BCALM\n");
        die("arguments: unitigs.fa");
    }
    std::map<long, long> count, totsize;
    for (const Record& r : read_fasta(argv[0])) {
        size_t p = 0;
        while (p < r.header.size()) {
            const size_t a = r.header.find_first_not_of(WS, p);
            if (a == std::string::npos) break;
            size_t b = r.header.find_first_of(WS, a);
            if (b == std::string::npos) b = r.header.size();
            const std::string field = r.header.substr(a, b - a);
            p = b;
            if (field.compare(0, 5, "km:f:") != 0) continue;
            const std::string v = field.substr(field.rfind(':') + 1);
            char* end = nullptr; const double d = strtod(v.c_str(), &end);
            if (end == v.c_str() || *end) die("could not convert string to float: '" + v + "'");
            const long ab = (long)d;                                           // int(float(x)): toward zero
            ++count[ab]; totsize[ab] += (long)r.seq.size();
        }
    }
    printf("'value' : 'number of unitigs having this mean abundance value' : 'total size of unitigs having this mean abundance'\n");
    for (const auto& kv : count) printf("%ld : %ld : %ld\n", kv.first, kv.second, totsize[kv.first]);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    const std::string cmd = argc > 1 ? argv[1] : "";
    if (cmd == "convertToGFA") return to_gfa(argc - 2, argv + 2);
    if (cmd == "split_unitigs") return split_main(false, argc - 2, argv + 2);
    if (cmd == "pufferize") return split_main(true, argc - 2, argv + 2);
    if (cmd == "abundance_stats") return abundance_main(argc - 2, argv + 2);
    fprintf(stderr, "usage: bcalm_tools <convertToGFA|split_unitigs|pufferize|abundance_stats> args...\n");
    return 1;
}
