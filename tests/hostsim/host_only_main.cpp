// ASan/UBSan driver for the library's pure-host C++ (mirge3.0_amd/csrc/native_host.hpp): the mapped.csv / unmapped.csv
// formatter, the GFF3 writer, the text of a merged library and the argument checks of mirge_lib_create_packed.
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -std=c++17 -pthread host_only_main.cpp -o host_only && ./host_only <dir>
// The GPU pool has no device sanitizer; these functions never touch the device, so the very translation unit the product
// compiles runs here under the host sanitizers.  Cases: names that need CSV quoting (comma, quote, line break), empty libraries
// and empty references, reads of 0, 1, 255 and 600 nt, 1 and 17 samples, unmapped-only and mapped-only outputs, row orders that are
// permutations, out-of-range indices that must be refused, a GFF with reads that hold an N and records at the text limit.
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mirge_native.h"
#include "../../mirge3.0_amd/csrc/mirge_core.hpp"
#include "../../mirge3.0_amd/csrc/mirge_isotype.hpp"
#include "../../mirge3.0_amd/csrc/mirge_libbuild.hpp"
#include "../../mirge3.0_amd/csrc/native_host.hpp"
#include <atomic>
#include "../../mirge3.0_amd/csrc/native_gz.hpp"

static std::string slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

#define EXPECT(cond)                                                                   \
    do {                                                                               \
        if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

// the formatter's output, restated with std::string (slow and obviously right)
static std::string quote(const std::string& s) {
    if (s.find_first_of(",\"\n\r") == std::string::npos) return s;
    std::string o = "\"";
    for (char c : s) { if (c == '"') o += '"'; o += c; }
    return o + "\"";
}

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    std::mt19937_64 rng(20260104);
    auto rnd_seq = [&](size_t L) { std::string s(L, 'A'); for (auto& c : s) c = "ACGTN"[rng() % 100 == 0 ? 4 : rng() % 4]; return s; };

    // ---------------- mirge_annotation_csv
    for (int S : {1, 17}) {
        const int n_pass = 10, n_cols = 9;
        int32_t col_of_pass[n_pass] = {0, 1, 2, 3, 4, 5, 6, 7, 8, -1};  // the last pass writes no column
        std::vector<std::vector<std::string>> names(n_pass);
        const char* awkward[] = {"plain", "with,comma", "with \"quote\"", "line\nbreak", "cr\rhere", "", "hsa-miR-1/2", ",", "\"\"", "x"};
        for (int p = 0; p < n_pass; p++) {
            const int nn = p == 3 ? 0 : 5 + p;  // pass 3: an empty library
            for (int k = 0; k < nn; k++) names[p].push_back(std::string(awkward[(p + k) % 10]) + (k % 3 ? "_" + std::to_string(k) : ""));
        }
        std::vector<std::string> blob(n_pass);
        std::vector<std::vector<int64_t>> off(n_pass);
        const char* nd[n_pass]; const int64_t* no[n_pass]; int64_t nn[n_pass];
        for (int p = 0; p < n_pass; p++) {
            off[p].push_back(0);
            for (auto& s : names[p]) { blob[p] += s; off[p].push_back((int64_t)blob[p].size()); }
            nd[p] = names[p].empty() ? nullptr : blob[p].data(); no[p] = off[p].data(); nn[p] = (int64_t)names[p].size();
        }
        const int64_t U = 40000;
        std::string seq; std::vector<int64_t> soff{0};
        std::vector<int8_t> pass(U); std::vector<int32_t> ref(U); std::vector<uint32_t> cnt((size_t)U * S);
        const size_t lens[] = {0, 1, 16, 22, 31, 32, 64, 255, 256, 600};
        for (int64_t i = 0; i < U; i++) {
            seq += rnd_seq(i < 2000 ? lens[i % 10] : 16 + rng() % 20); soff.push_back((int64_t)seq.size());
            int p = (int)(rng() % (n_pass + 3)) - 3;
            if (p == 3) p = -1;  // nothing can be annotated to the empty library
            pass[i] = (int8_t)(p < 0 ? -1 : p);
            ref[i] = p >= 0 ? (int32_t)(rng() % names[p].size()) : -1;
            for (int s = 0; s < S; s++) cnt[(size_t)i * S + s] = (uint32_t)(rng() % 7 == 0 ? 4000000000u + rng() % 1000 : rng() % 1000);
        }
        std::vector<int64_t> rows(U);
        for (int64_t i = 0; i < U; i++) rows[i] = i;
        std::shuffle(rows.begin(), rows.end(), rng);
        std::string header = "Sequence,annotFlag";
        for (int c = 0; c < n_cols; c++) header += ",col" + std::to_string(c);
        for (int s = 0; s < S; s++) header += ",S" + std::to_string(s);
        header += "\n";
        const std::string pm = dir + "/m.csv", pu = dir + "/u.csv";
        int rc = mirge_annotation_csv(pm.c_str(), pu.c_str(), header.c_str(), seq.data(), soff.data(), pass.data(), ref.data(), cnt.data(), S,
                                      rows.data(), U, n_pass, col_of_pass, n_cols, nd, no, nn);
        EXPECT(rc == 0);
        std::string wm = header, wu = header;
        for (int64_t k = 0; k < U; k++) {
            const int64_t i = rows[k];
            std::string& o = pass[i] >= 0 ? wm : wu;
            o += quote(seq.substr((size_t)soff[i], (size_t)(soff[i + 1] - soff[i]))) + "," + (pass[i] >= 0 ? "1" : "0");
            for (int c = 0; c < n_cols; c++) { o += ","; if (pass[i] >= 0 && col_of_pass[pass[i]] == c) o += quote(names[pass[i]][ref[i]]); }
            for (int s = 0; s < S; s++) o += "," + std::to_string(cnt[(size_t)i * S + s]);
            o += "\n";
        }
        EXPECT(slurp(pm) == wm);
        EXPECT(slurp(pu) == wu);
        // one file only, no rows at all, and what must be refused
        EXPECT(mirge_annotation_csv(nullptr, pu.c_str(), header.c_str(), seq.data(), soff.data(), pass.data(), ref.data(), cnt.data(), S, rows.data(), U,
                                    n_pass, col_of_pass, n_cols, nd, no, nn) == 0 && slurp(pu) == wu);
        EXPECT(mirge_annotation_csv(pm.c_str(), nullptr, header.c_str(), seq.data(), soff.data(), pass.data(), ref.data(), cnt.data(), S, rows.data(), 0,
                                    n_pass, col_of_pass, n_cols, nd, no, nn) == 0 && slurp(pm) == header);
        ref[rows[5]] = 1 << 20; pass[rows[5]] = 0;
        EXPECT(mirge_annotation_csv(pm.c_str(), pu.c_str(), header.c_str(), seq.data(), soff.data(), pass.data(), ref.data(), cnt.data(), S, rows.data(), U,
                                    n_pass, col_of_pass, n_cols, nd, no, nn) == -1);
        ref[rows[5]] = 0; pass[rows[5]] = 3;  // the empty library
        EXPECT(mirge_annotation_csv(pm.c_str(), pu.c_str(), header.c_str(), seq.data(), soff.data(), pass.data(), ref.data(), cnt.data(), S, rows.data(), U,
                                    n_pass, col_of_pass, n_cols, nd, no, nn) == -1);
        pass[rows[5]] = 11;
        EXPECT(mirge_annotation_csv(pm.c_str(), pu.c_str(), header.c_str(), seq.data(), soff.data(), pass.data(), ref.data(), cnt.data(), S, rows.data(), U,
                                    n_pass, col_of_pass, n_cols, nd, no, nn) == -1);
        EXPECT(std::strlen(mirge_last_error()) > 0);
    }

    // ---------------- mirge_gff_write
    {
        const int64_t n = 30000;
        const int S = 3;
        std::vector<MirgeIsoRec> rec((size_t)n);
        std::string reads; std::vector<int64_t> roff{0};
        std::vector<uint32_t> cnt((size_t)n * S);
        std::vector<int32_t> name_of(n), parent_of(n);
        std::string nblob, pblob; std::vector<int64_t> noff{0}, poff{0};
        for (int k = 0; k < 50; k++) { nblob += "hsa-miR-" + std::to_string(k) + (k % 7 ? "-5p" : ""); noff.push_back((int64_t)nblob.size()); }
        for (int k = 0; k < 20; k++) { pblob += k ? "hsa-mir-" + std::to_string(k) : ""; poff.push_back((int64_t)pblob.size()); }
        for (int64_t k = 0; k < n; k++) {
            std::memset(&rec[(size_t)k], 0, sizeof(MirgeIsoRec));
            rec[(size_t)k].kind = (int32_t)(rng() % 3);
            rec[(size_t)k].start = (int32_t)(rng() % 100); rec[(size_t)k].end = rec[(size_t)k].start + 22;
            const int vl = k % 97 == 0 ? 200 : (int)(rng() % 40), cl = k % 97 == 0 ? MIRGE_ISO_TEXT - 200 : (int)(rng() % 30);  // some at the text limit
            rec[(size_t)k].vlen = vl; rec[(size_t)k].clen = cl;
            for (int q = 0; q < vl + cl; q++) rec[(size_t)k].text[q] = "iso_snv_3pMI0-9:,"[rng() % 17];
            reads += rnd_seq(k % 211 == 0 ? 0 : 15 + rng() % 15); roff.push_back((int64_t)reads.size());
            for (int s = 0; s < S; s++) cnt[(size_t)k * S + s] = (uint32_t)(rng() % 100000);
            name_of[k] = (int32_t)(rng() % 50); parent_of[k] = (int32_t)(rng() % 20);
        }
        const std::string pg = dir + "/s.gff";
        EXPECT(mirge_gff_write(pg.c_str(), "## head\n", "miRBase22", rec.data(), n, reads.data(), roff.data(), cnt.data(), S, name_of.data(), nblob.data(),
                               noff.data(), 50, parent_of.data(), pblob.data(), poff.data(), 20, nullptr, 0) == 0);
        const std::string g = slurp(pg);
        int64_t lines = 0, want = 0;
        for (char c : g) lines += c == '\n';
        for (int64_t k = 0; k < n; k++) want += rec[(size_t)k].kind != 0;
        EXPECT(lines == want + 1 && g.rfind("## head\n", 0) == 0);
        {   // the rows' reads and counts through an index into a larger table (read_of_row): the same file; an index beyond it is refused
            std::vector<int64_t> perm((size_t)n);
            for (int64_t k = 0; k < n; k++) perm[(size_t)k] = n - 1 - k;
            std::string reads2; std::vector<int64_t> roff2{0};
            std::vector<uint32_t> cnt2((size_t)n * S);
            for (int64_t j = 0; j < n; j++) {  // table entry j = row n-1-j
                const int64_t k = n - 1 - j;
                reads2.append(reads, (size_t)roff[(size_t)k], (size_t)(roff[(size_t)k + 1] - roff[(size_t)k])); roff2.push_back((int64_t)reads2.size());
                for (int s2 = 0; s2 < S; s2++) cnt2[(size_t)j * S + s2] = cnt[(size_t)k * S + s2];
            }
            const std::string pg2 = dir + "/s2.gff";
            EXPECT(mirge_gff_write(pg2.c_str(), "## head\n", "miRBase22", rec.data(), n, reads2.data(), roff2.data(), cnt2.data(), S, name_of.data(), nblob.data(),
                                   noff.data(), 50, parent_of.data(), pblob.data(), poff.data(), 20, perm.data(), n) == 0);
            EXPECT(slurp(pg2) == g);
            perm[3] = n; rec[3].kind = 1;
            EXPECT(mirge_gff_write(pg2.c_str(), "## head\n", "miRBase22", rec.data(), n, reads2.data(), roff2.data(), cnt2.data(), S, name_of.data(), nblob.data(),
                                   noff.data(), 50, parent_of.data(), pblob.data(), poff.data(), 20, perm.data(), n) == -1);
        }
        EXPECT(mirge_gff_write(pg.c_str(), "## head\n", "x", rec.data(), 0, nullptr, roff.data(), nullptr, S, nullptr, nblob.data(), noff.data(), 50, nullptr,
                               pblob.data(), poff.data(), 20, nullptr, 0) == 0 && slurp(pg) == "## head\n");
        name_of[7] = 50; rec[7].kind = 1;
        EXPECT(mirge_gff_write(pg.c_str(), "## head\n", "x", rec.data(), n, reads.data(), roff.data(), cnt.data(), S, name_of.data(), nblob.data(), noff.data(),
                               50, parent_of.data(), pblob.data(), poff.data(), 20, nullptr, 0) == -1);
        name_of[7] = 0; rec[7].vlen = MIRGE_ISO_TEXT; rec[7].clen = 1;
        EXPECT(mirge_gff_write(pg.c_str(), "## head\n", "x", rec.data(), n, reads.data(), roff.data(), cnt.data(), S, name_of.data(), nblob.data(), noff.data(),
                               50, parent_of.data(), pblob.data(), poff.data(), 20, nullptr, 0) == -1);
    }

    // ---------------- merged_library_text: the members' letters back out of their packed images, N where the image says invalid
    {
        std::vector<MirgeHostLib> hl(3);
        std::vector<std::string> seqs[3];
        std::string all; std::vector<int64_t> alloff{0};
        for (int m = 0; m < 3; m++) {
            std::string s; std::vector<int64_t> off{0};
            const int nr = m == 1 ? 0 : 40;  // an empty member library
            for (int r = 0; r < nr; r++) {
                const std::string q = r % 9 == 0 ? std::string() : rnd_seq(1 + rng() % 300);  // empty references too
                seqs[m].push_back(q); s += q; off.push_back((int64_t)s.size());
                all += q; alloff.push_back((int64_t)all.size());
            }
            std::string err;
            EXPECT(mirge_hostlib_build(hl[(size_t)m], s.data(), off.data(), nr, err) == 0);
        }
        const MirgeHostLib* members[3] = {&hl[0], &hl[1], &hl[2]};
        std::string seq; std::vector<int64_t> off;
        merged_library_text(members, 3, seq, off);
        EXPECT(seq == all && off == alloff);
        merged_library_text(members + 1, 1, seq, off);
        EXPECT(seq.empty() && off.size() == 1);
        // ... and the packed image goes through mirge_lib_create_packed's checks; damaged ones do not
        const MirgeHostLib& h = hl[0];
        EXPECT(lib_packed_args_check(h.T.data(), (int64_t)h.T.size(), h.inv.data(), (int64_t)h.inv.size(), h.ref_start.data(), h.n_refs, h.total, h.kmax) == 0);
        std::vector<uint32_t> rs = h.ref_start;
        std::swap(rs[3], rs[4]);
        EXPECT(lib_packed_args_check(h.T.data(), (int64_t)h.T.size(), h.inv.data(), (int64_t)h.inv.size(), rs.data(), h.n_refs, h.total, h.kmax) == -1);
        rs = h.ref_start; rs[0] = 1;
        EXPECT(lib_packed_args_check(h.T.data(), (int64_t)h.T.size(), h.inv.data(), (int64_t)h.inv.size(), rs.data(), h.n_refs, h.total, h.kmax) == -1);
        EXPECT(lib_packed_args_check(h.T.data(), (int64_t)h.T.size() - 1, h.inv.data(), (int64_t)h.inv.size(), h.ref_start.data(), h.n_refs, h.total, h.kmax) == -1);
        EXPECT(lib_packed_args_check(h.T.data(), (int64_t)h.T.size(), h.inv.data(), (int64_t)h.inv.size(), h.ref_start.data(), h.n_refs, h.total + 1, h.kmax) == -1);
        EXPECT(lib_packed_args_check(h.T.data(), (int64_t)h.T.size(), h.inv.data(), (int64_t)h.inv.size(), h.ref_start.data(), h.n_refs, h.total, 7) == -1);
        EXPECT(lib_packed_args_check(nullptr, 8, h.inv.data(), 4, h.ref_start.data(), 0, 0, 8) == -1);
    }
    // ---------------- mirge_gz_inflate: gzip members cut at block starts found by search, decoded without their history, resolved
    {
        auto gz_compress = [&](const std::string& text, int level, int strategy, size_t flush_every) {
            z_stream zs;
            std::memset(&zs, 0, sizeof(zs));
            deflateInit2(&zs, level, Z_DEFLATED, 31, 8, strategy);
            std::string out(compressBound((uLong)text.size()) + text.size() / 100 + 4096, '\0');
            zs.next_out = (Bytef*)out.data(); zs.avail_out = (uInt)out.size();
            size_t at = 0;
            while (at < text.size()) {
                const size_t m = std::min(text.size() - at, flush_every ? flush_every : text.size());
                zs.next_in = (Bytef*)text.data() + at; zs.avail_in = (uInt)m;
                at += m;
                deflate(&zs, at == text.size() ? Z_FINISH : Z_SYNC_FLUSH);  // sync flushes: what pigz writes between its pieces
            }
            if (text.empty()) deflate(&zs, Z_FINISH);
            out.resize(zs.total_out);
            deflateEnd(&zs);
            return out;
        };
        auto fastq = [&](size_t n_rec, int qual_kinds) {
            std::string t;
            for (size_t i = 0; i < n_rec; i++) {
                const size_t L = 16 + rng() % 35;
                t += "@SRR" + std::to_string(1000000 + i) + " " + std::to_string(i) + " length=" + std::to_string(L) + "\n";
                std::string q(L, 'I');
                for (auto& c : q) c = (char)(33 + (qual_kinds <= 1 ? 40 : rng() % qual_kinds));
                t += (rng() % 3 ? std::string("TGAGGTAGTAGGTTGTATAGTT").substr(0, std::min<size_t>(L, 22)) + rnd_seq(L > 22 ? L - 22 : 0) : rnd_seq(L)) + "\n+\n" + q + "\n";
            }
            return t;
        };
        int taken = 0, refused = 0;
        const std::string texts[] = {fastq(150000, 1), fastq(120000, 41), std::string(3000000, 'A'), fastq(20, 4)};
        const int combos[][3] = {{6, 0, 8}, {1, 1 << 20, 3}, {9, 0, 2}, {6, 1 << 19, 8}, {6, 0, 1}};  // level, sync flush every, threads
        for (const std::string& text : texts)
            for (const auto& cb : combos) {
                const std::string gz = gz_compress(text, cb[0], Z_DEFAULT_STRATEGY, (size_t)cb[1]);
                std::vector<uint8_t> out(text.size() + 1);
                int64_t n = -1;
                const int rc = mirge_gz_inflate((const uint8_t*)gz.data(), (int64_t)gz.size(), out.data(), (int64_t)text.size(), &n, cb[2]);
                if (rc == 0) { taken++; EXPECT(n == (int64_t)text.size() && std::memcmp(out.data(), text.data(), text.size()) == 0); }
                else refused++;
                if (gz.size() > (3u << 20)) EXPECT(rc == 0);  // several MiB of compressed text: the route must take it
            }
        EXPECT(taken >= 8);
        // what it must not take for the truth: a flipped bit (CRC), a truncated file, a second member behind the first, too little room
        const std::string text = fastq(150000, 41), gz = gz_compress(text, 6, Z_DEFAULT_STRATEGY, 0);
        std::vector<uint8_t> out(text.size() + 16);
        int64_t n = 0;
        for (size_t where : {gz.size() / 3, gz.size() / 2, gz.size() - 20}) {
            std::string bad = gz;
            bad[where] ^= 0x20;
            EXPECT(mirge_gz_inflate((const uint8_t*)bad.data(), (int64_t)bad.size(), out.data(), (int64_t)out.size(), &n, 4) != 0);
        }
        EXPECT(mirge_gz_inflate((const uint8_t*)gz.data(), (int64_t)gz.size() / 2, out.data(), (int64_t)out.size(), &n, 4) != 0);
        // several members (lanes merged with cat): a large one, a small one (zlib on this thread), an empty one, zero padding
        const std::string small_t = fastq(300, 41), small_gz = gz_compress(small_t, 6, Z_DEFAULT_STRATEGY, 0), empty_gz = gz_compress(std::string(), 6, Z_DEFAULT_STRATEGY, 0);
        const std::string multi = gz + small_gz + empty_gz + gz + std::string(64, '\0'), multi_t = text + small_t + text;
        std::vector<uint8_t> out2(multi_t.size() + 1000);
        EXPECT(mirge_gz_inflate((const uint8_t*)multi.data(), (int64_t)multi.size(), out2.data(), (int64_t)out2.size(), &n, 4) == 0);
        EXPECT(n == (int64_t)multi_t.size() && std::memcmp(out2.data(), multi_t.data(), multi_t.size()) == 0);
        EXPECT(mirge_gz_inflate((const uint8_t*)multi.data(), (int64_t)multi.size(), out2.data(), (int64_t)multi_t.size() - 5, &n, 4) != 0);  // too little room
        // mirge_gz_inflate_progress: a second thread reads the prefix that is reported final WHILE the members inflate -- every
        // byte of it must already be the text's, the reports only grow, and the last one is the whole length
        {
            std::fill(out2.begin(), out2.end(), (uint8_t)0);
            int64_t progress = 0, seen_max = 0, checks = 0;
            std::atomic<int> stop{0}, wrong{0};
            std::thread reader([&]() {
                int64_t last = 0;
                while (true) {
                    const int last_round = stop.load();
                    const int64_t pr = __atomic_load_n(&progress, __ATOMIC_ACQUIRE);
                    if (pr < last || pr > (int64_t)multi_t.size()) wrong = 1;
                    if (pr > last) {
                        if (std::memcmp(out2.data() + last, multi_t.data() + last, (size_t)(pr - last)) != 0) wrong = 1;
                        last = pr;
                        checks++;
                    }
                    if (last_round) break;
                }
                seen_max = last;
            });
            const int rc = mirge_gz_inflate_progress((const uint8_t*)multi.data(), (int64_t)multi.size(), out2.data(), (int64_t)out2.size(), &n, 4, &progress);
            stop = 1;
            reader.join();
            EXPECT(rc == 0 && !wrong && seen_max == (int64_t)multi_t.size() && n == seen_max && checks >= 3);
        }
        std::string junk = gz + "garbage behind the member";
        EXPECT(mirge_gz_inflate((const uint8_t*)junk.data(), (int64_t)junk.size(), out2.data(), (int64_t)out2.size(), &n, 4) != 0);
        std::string badsmall = gz + small_gz;
        badsmall[gz.size() + small_gz.size() / 2] ^= 0x10;
        EXPECT(mirge_gz_inflate((const uint8_t*)badsmall.data(), (int64_t)badsmall.size(), out2.data(), (int64_t)out2.size(), &n, 4) != 0);
        EXPECT(mirge_gz_inflate((const uint8_t*)gz.data(), (int64_t)gz.size(), out.data(), (int64_t)text.size() - 1, &n, 4) != 0);
        EXPECT(mirge_gz_inflate((const uint8_t*)"not a gzip file at all, no", 26, out.data(), 100, &n, 4) != 0);
        // BGZF: members of <= 64 KiB that carry their size in an extra field, an empty one at the end
        std::string bg;
        for (size_t at = 0; at <= text.size(); at += 60000) {
            const std::string piece = at < text.size() ? text.substr(at, 60000) : std::string();
            z_stream zs;
            std::memset(&zs, 0, sizeof(zs));
            deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
            std::string body(compressBound((uLong)piece.size()) + 64, '\0');
            zs.next_in = (Bytef*)piece.data(); zs.avail_in = (uInt)piece.size();
            zs.next_out = (Bytef*)body.data(); zs.avail_out = (uInt)body.size();
            deflate(&zs, Z_FINISH);
            body.resize(zs.total_out);
            deflateEnd(&zs);
            const uint32_t crc = (uint32_t)crc32(0L, (const Bytef*)piece.data(), (uInt)piece.size()), isz = (uint32_t)piece.size();
            const uint16_t bsize = (uint16_t)(18 + body.size() + 8 - 1);
            const unsigned char hd[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (unsigned char)(bsize & 255), (unsigned char)(bsize >> 8)};
            bg.append((const char*)hd, 18); bg += body;
            for (int k = 0; k < 4; k++) bg.push_back((char)((crc >> (8 * k)) & 255));
            for (int k = 0; k < 4; k++) bg.push_back((char)((isz >> (8 * k)) & 255));
        }
        std::vector<uint8_t> out3(text.size() + 65536);
        EXPECT(mirge_gz_inflate((const uint8_t*)bg.data(), (int64_t)bg.size(), out3.data(), (int64_t)out3.size(), &n, 5) == 0);
        EXPECT(n == (int64_t)text.size() && std::memcmp(out3.data(), text.data(), text.size()) == 0);
        bg[bg.size() / 2] ^= 1;
        EXPECT(mirge_gz_inflate((const uint8_t*)bg.data(), (int64_t)bg.size(), out3.data(), (int64_t)out3.size(), &n, 5) != 0);
        std::printf("gz: %d inflated in parallel and equal, %d left to the serial route\n", taken, refused);
    }
    std::printf("host-only functions clean\n");
    return 0;
}
