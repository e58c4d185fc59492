// ASan/UBSan driver for the shared device/host arithmetic (mirge_core.hpp, mirge_libbuild.hpp):
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -std=c++17 sanitize_main.cpp -o _build/sanitize && _build/sanitize
// GPU AddressSanitizer is not available on the test pool, so the sanitizers run on this CPU build.
#include <cstdio>
#include <random>
#include <string>
#include <vector>

#include "hostsim.cpp"

int main() {
    std::mt19937_64 rng(12345);
    auto rnd_seq = [&](int L, double pn) {
        std::string s(L, 'A');
        for (auto& c : s) c = (rng() % 1000 < pn * 1000) ? 'N' : "ACGT"[rng() % 4];
        return s;
    };
    const int n_pass = 9;
    std::vector<std::string> libseq(n_pass);
    std::vector<std::vector<int64_t>> liboff(n_pass);
    const int nref[n_pass] = {300, 200, 50, 50, 80, 4, 300, 200, 300};
    const int lmin[n_pass] = {18, 60, 70, 90, 60, 120, 100, 500, 18}, lmax[n_pass] = {25, 120, 90, 120, 250, 3000, 1000, 3000, 25};
    for (int p = 0; p < n_pass; p++) {
        liboff[p].push_back(0);
        for (int r = 0; r < nref[p]; r++) {
            libseq[p] += rnd_seq(lmin[p] + (int)(rng() % (lmax[p] - lmin[p] + 1)), p == 6 ? 0.002 : 0.0);
            liboff[p].push_back((int64_t)libseq[p].size());
        }
    }
    libseq[8] = libseq[0]; liboff[8] = liboff[0];
    std::string reads; std::vector<int64_t> roff{0};
    for (int i = 0; i < 20000; i++) {
        int L = 1 + (int)(rng() % (i % 50 == 0 ? 128 : 40));
        std::string s;
        int p = (int)(rng() % n_pass);
        int r = (int)(rng() % nref[p]);
        int64_t a = liboff[p][r], b = liboff[p][r + 1];
        if (rng() % 4 && b - a > L) { int64_t o = a + (int64_t)(rng() % (uint64_t)(b - a - L)); s = libseq[p].substr((size_t)o, (size_t)L); }
        else s = rnd_seq(L, 0.01);
        for (int m = (int)(rng() % 3); m > 0; m--) s[rng() % s.size()] = "ACGTN"[rng() % 5];
        if (rng() % 20 == 0) s += std::string(3 + rng() % 4, 'T');
        if ((int)s.size() > 128) s.resize(128);
        reads += s; roff.push_back((int64_t)reads.size());
    }
    const int64_t n = (int64_t)roff.size() - 1;
    MirgePolicy pol[n_pass] = {
        {0, 0, 28, 2, 0, 0, 0, 26, 0, 0}, {0, 1, 28, 2, 0, 0, 0, 0, 25, 0}, {1, 1, 28, 1, 0, 0, 0, 0, 0, 0},
        {1, 0, 28, 0, 0, 0, 1, 0, 0, 0}, {0, 1, 28, 2, 0, 0, 0, 0, 0, 0}, {0, 1, 28, 2, 0, 0, 0, 0, 0, 0},
        {0, 1, 28, 2, 0, 0, 0, 0, 0, 0}, {0, 0, 28, 2, 0, 0, 0, 0, 0, 0}, {1, 2, 28, 2, 1, 2, 0, 0, 0, 0}};
    const char* ls[n_pass]; const int64_t* lo[n_pass]; int64_t ln[n_pass];
    for (int p = 0; p < n_pass; p++) { ls[p] = libseq[p].data(); lo[p] = liboff[p].data(); ln[p] = nref[p]; }
    std::vector<int8_t> ps(n), mm(n); std::vector<int32_t> ref(n), off(n);
    int rc = hostsim_cascade(reads.data(), roff.data(), n, ls, lo, ln, pol, n_pass, ps.data(), ref.data(), off.data(), mm.data());
    long hits = 0; unsigned long long sum = 0;
    for (int64_t i = 0; i < n; i++) if (ps[i] >= 0) { hits++; sum += (unsigned long long)ps[i] * 131 + (unsigned)ref[i] * 7 + (unsigned)off[i]; }
    std::printf("rc=%d reads=%ld annotated=%ld checksum=%llu\n", rc, (long)n, hits, sum);
    // the isomiR typing, both forms (mirge_isotype on arrays, mirge_isotype_fast on bit planes: shifts by run-time amounts)
    int64_t bad = 0;
    char fa[100], fb[200], fp[300];
    const int64_t took = hostsim_isotype_fuzz(7, 150000, &bad, fa, fb, fp);
    std::printf("isotype pairs=%ld differing=%ld\n", (long)took, (long)bad);
    return rc ? rc : (bad != 0 || took < 100000);
}
