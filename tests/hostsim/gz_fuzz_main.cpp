// Test infrastructure: mutation fuzz of mirge_gz_inflate (mirge3.0_amd/csrc/native_gz.hpp), built with -fsanitize=address,undefined.
// A .gz sample (FASTQ-like text, ordinary member / pigz-style sync flushes / two members / BGZF) is damaged the ways a download or a
// disk damages files -- flipped bits, overwritten bytes, zeroed or duplicated spans, a cut end -- and handed to the inflater on 1-6
// threads.  Required of every mutant: no sanitizer report, no hang, and IF the inflater says 0 the text is byte for byte what zlib
// makes of the same bytes (zlib accepting is then required too).  A refusal is always acceptable: the caller's streamed zlib route
// takes over and reports the damage.
//   gz_fuzz <mutants> <seed>
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
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
#include "../../mirge3.0_amd/csrc/native_gz.hpp"

static std::mt19937_64 rng;

static std::string gz_compress(const std::string& text, int level, size_t flush_every) {
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    deflateInit2(&zs, level, Z_DEFLATED, 31, 8, Z_DEFAULT_STRATEGY);
    std::string out(compressBound((uLong)text.size()) + text.size() / 100 + 4096, '\0');
    zs.next_out = (Bytef*)out.data(); zs.avail_out = (uInt)out.size();
    size_t at = 0;
    while (at < text.size()) {
        const size_t m = std::min(text.size() - at, flush_every ? flush_every : text.size());
        zs.next_in = (Bytef*)text.data() + at; zs.avail_in = (uInt)m;
        at += m;
        deflate(&zs, at == text.size() ? Z_FINISH : Z_SYNC_FLUSH);
    }
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return out;
}

static std::string bgzf_compress(const std::string& text) {
    std::string out;
    for (size_t at = 0; at <= text.size(); at += 65280) {  // the last round writes the empty end-of-file block
        const size_t m = at < text.size() ? std::min<size_t>(65280, text.size() - at) : 0;
        z_stream zs;
        std::memset(&zs, 0, sizeof(zs));
        deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        std::string body(compressBound((uLong)m) + 64, '\0');
        zs.next_in = (Bytef*)text.data() + std::min(at, text.size()); zs.avail_in = (uInt)m;
        zs.next_out = (Bytef*)body.data(); zs.avail_out = (uInt)body.size();
        deflate(&zs, Z_FINISH);
        body.resize(zs.total_out);
        deflateEnd(&zs);
        const uint32_t crc = (uint32_t)crc32(0, (const Bytef*)text.data() + std::min(at, text.size()), (uInt)m), isz = (uint32_t)m;
        const uint16_t bsize = (uint16_t)(18 + body.size() + 8 - 1);
        const unsigned char head[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, (unsigned char)(bsize & 255), (unsigned char)(bsize >> 8)};
        out.append((const char*)head, 18);
        out += body;
        for (int k = 0; k < 4; k++) out.push_back((char)((crc >> (8 * k)) & 255));
        for (int k = 0; k < 4; k++) out.push_back((char)((isz >> (8 * k)) & 255));
        if (at >= text.size()) break;
    }
    return out;
}

// what zlib (gzip's rules: members back to back, zero padding behind them) makes of the bytes; false = zlib refuses
static bool zlib_inflate(const std::string& gz, std::string& text, size_t cap) {
    text.clear();
    size_t at = 0;
    bool any = false;
    while (at < gz.size()) {
        if (any && gz[at] == 0) { at++; continue; }
        z_stream zs;
        std::memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, 31) != Z_OK) return false;
        zs.next_in = (Bytef*)gz.data() + at; zs.avail_in = (uInt)(gz.size() - at);
        int rc = Z_OK;
        while (rc == Z_OK) {
            const size_t old = text.size();
            if (old >= cap) { inflateEnd(&zs); return false; }
            text.resize(std::min(cap, old + (1u << 20)));
            zs.next_out = (Bytef*)&text[old]; zs.avail_out = (uInt)(text.size() - old);
            rc = inflate(&zs, Z_NO_FLUSH);
            text.resize(text.size() - zs.avail_out);
        }
        const size_t used = gz.size() - at - zs.avail_in;
        inflateEnd(&zs);
        if (rc != Z_STREAM_END) return false;
        at += used;
        any = true;
    }
    return any;
}

int main(int argc, char** argv) {
    const int mutants = argc > 1 ? std::atoi(argv[1]) : 200;
    rng.seed(argc > 2 ? (uint64_t)std::atoll(argv[2]) : 1);
    // ~6.5 MB of FASTQ-like text -> ~1.3 MB at level 6: two chunks of the parallel route (>= 512 KiB of compressed data each)
    std::string text;
    {
        const char* B = "ACGT";
        for (size_t i = 0; i < 75000; i++) {
            const size_t L = 16 + rng() % 35;
            text += "@SRR" + std::to_string(1000000 + i) + " " + std::to_string(i) + " length=" + std::to_string(L) + "\n";
            std::string s(L, 'A'), q(L, 'I');
            for (size_t k = 0; k < L; k++) { s[k] = rng() % 3 && k < 22 ? "TGAGGTAGTAGGTTGTATAGTT"[k] : B[rng() & 3]; q[k] = (char)(33 + rng() % 41); }
            text += s + "\n+\n" + q + "\n";
        }
    }
    const std::string half = text.substr(0, text.size() / 2);
    const std::string samples[4] = {gz_compress(text, 6, 0), gz_compress(text, 1, 1 << 20), gz_compress(half, 6, 0) + gz_compress(text, 6, 0), bgzf_compress(half)};
    std::vector<uint8_t> out(text.size() * 2 + (1 << 20));
    std::string ref;
    int accepted = 0, refused = 0, zlib_ok_refused = 0;
    for (int m = 0; m < mutants; m++) {
        std::string gz = samples[m % 4];
        const int kind = (int)(rng() % 7), reps = 1 + (int)(rng() % 3);
        for (int r = 0; r < reps; r++) {
            const size_t at = rng() % gz.size();
            switch (kind) {
                case 0: gz[at] ^= (char)(1 << (rng() % 8)); break;                                   // one bit
                case 1: gz[at] = (char)rng(); break;                                                  // one byte
                case 2: { const size_t n = std::min<size_t>(gz.size() - at, 1 + rng() % 4096); std::memset(&gz[at], 0, n); break; }  // a zeroed span
                case 3: { const size_t n = std::min<size_t>(gz.size() - at, 1 + rng() % 4096); gz.insert(at, gz.substr(at, n)); break; }  // a repeated span
                case 4: gz.resize(std::max<size_t>(18, at)); break;                                    // a cut end
                case 5: { const size_t n = std::min<size_t>(gz.size() - at, 1 + rng() % 64); for (size_t k = 0; k < n; k++) gz[at + k] = (char)rng(); break; }  // noise
                default: if (m % 5 == 0 && r == 0) gz[rng() % std::min<size_t>(gz.size(), 32)] ^= (char)(1 << (rng() % 8)); break;  // the header / nothing
            }
        }
        int64_t n = -1;
        const int threads = 1 + (int)(rng() % 6);
        const int rc = mirge_gz_inflate((const uint8_t*)gz.data(), (int64_t)gz.size(), out.data(), (int64_t)out.size(), &n, threads);
        const bool z = zlib_inflate(gz, ref, out.size());
        if (rc == 0) {
            accepted++;
            if (!z || n != (int64_t)ref.size() || std::memcmp(out.data(), ref.data(), ref.size()) != 0) {
                std::printf("FAIL mutant %d (sample %d, kind %d, %d threads): accepted %lld bytes, zlib %s %zu\n", m, m % 4, kind, threads, (long long)n, z ? "gives" : "refuses after", ref.size());
                return 1;
            }
        } else {
            refused++;
            zlib_ok_refused += z;
        }
    }
    std::printf("gz fuzz clean: %d mutants, %d inflated in parallel and equal to zlib, %d refused (%d of them fine for zlib: the serial route's)\n", mutants, accepted, refused, zlib_ok_refused);
    return 0;
}
