// kernels_csv.hpp -- part of mirge_kernels.hpp: the text of mapped.csv / unmapped.csv (mirge/__main__.py:164-173) formatted
// on the device from what already lives there -- packed unique reads, their count matrix, the cascade's (pass, reference)
// -- instead of fetching 35 B per unique read to the host and formatting 200 MB with its cores (round 2: 0.15 of the 0.20 s a
// resident 10 M-read sample took).
//   k_csv_rowlen : bytes of every row, into one of two length arrays (mapped | unmapped)     -> two exclusive scans
//   k_csv_rows   : every row's bytes at its offset: sequence, annotFlag, the name in its pass's column, the S counts
// A row:  SEQ,<0|1>,<name col 0>,...,<name col n-1>,<count 0>,...,<count S-1>\n  -- the host's mirge_annotation_csv byte
// for byte; names that would need CSV quoting (a comma, a quote, a line break) are excluded by the caller, which then takes
// the host route.
#pragma once

#define MIRGE_CSV_MAXG MIRGE_NCLS
struct CsvGroup {
    const uint64_t* seq;     // [W][n]
    const uint64_t* nmask;   // [W][n] or nullptr
    const uint8_t* len;      // [n]
    const uint32_t* counts;  // [n][S]
    const int8_t* pass;      // [n]
    const int32_t* ref;      // [n]
    uint32_t base, n;
    int32_t W;
    int32_t len16;           // 1: `len` holds uint16 (the long read class)
};
__device__ __forceinline__ int csv_len(const CsvGroup& g, uint32_t j) {
    return g.len16 ? (int)reinterpret_cast<const uint16_t*>(g.len)[j] : (int)g.len[j];
}
struct CsvTables {
    CsvGroup g[MIRGE_CSV_MAXG];
    const uint8_t* name_data[MIRGE_MAX_PASSES_K];   // all names of pass p, back to back
    const uint32_t* name_off[MIRGE_MAX_PASSES_K];   // [n_names + 1]
    uint32_t name_n[MIRGE_MAX_PASSES_K];
    int32_t col_of_pass[MIRGE_MAX_PASSES_K];
    int32_t n_pass, n_name_cols, S;
};

__device__ __forceinline__ int csv_digits(uint32_t v) {
    return v < 10u ? 1 : v < 100u ? 2 : v < 1000u ? 3 : v < 10000u ? 4 : v < 100000u ? 5 : v < 1000000u ? 6 : v < 10000000u ? 7
         : v < 100000000u ? 8 : v < 1000000000u ? 9 : 10;
}

// the group that holds handle index i (groups are consecutive in handle order) and the read's slot in it
__device__ __forceinline__ int csv_locate(const CsvTables& t, uint32_t i, uint32_t& j) {
    int gi = 0;
#pragma unroll
    for (int k = 1; k < MIRGE_CSV_MAXG; k++)
        if (t.g[k].n && i >= t.g[k].base) gi = k;
    j = i - t.g[gi].base;
    return gi;
}

// flags[0] |= 1: a pass or reference index out of range
__global__ void k_csv_rowlen(CsvTables t, const uint32_t* __restrict__ rows, uint32_t n_rows, unsigned long long* __restrict__ len_m,
                             unsigned long long* __restrict__ len_u, uint32_t* __restrict__ flags) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_rows; k += gridDim.x * blockDim.x) {
        uint32_t j;
        const int gi = csv_locate(t, rows[k], j);
        const CsvGroup& g = t.g[gi];
        const int p = g.pass[j];
        uint32_t nlen = 0;
        if (p >= t.n_pass) { atomicOr(&flags[0], 1u); len_m[k] = 0; len_u[k] = 0; continue; }
        if (p >= 0 && t.col_of_pass[p] >= 0) {
            const int32_t r = g.ref[j];
            if (!t.name_off[p] || r < 0 || (uint32_t)r >= t.name_n[p]) { atomicOr(&flags[0], 1u); len_m[k] = 0; len_u[k] = 0; continue; }
            nlen = t.name_off[p][r + 1] - t.name_off[p][r];
        }
        uint32_t total = (uint32_t)csv_len(g, j) + 2u + (uint32_t)t.n_name_cols + nlen + (uint32_t)t.S + 1u;
        for (int s = 0; s < t.S; s++) total += (uint32_t)csv_digits(g.counts[(size_t)j * t.S + s]);
        len_m[k] = p >= 0 ? total : 0ull;
        len_u[k] = p >= 0 ? 0ull : total;
    }
}

__global__ void k_csv_rows(CsvTables t, const uint32_t* __restrict__ rows, uint32_t n_rows, const unsigned long long* __restrict__ off_m,
                           const unsigned long long* __restrict__ off_u, uint8_t* __restrict__ out_m, uint8_t* __restrict__ out_u) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_rows; k += gridDim.x * blockDim.x) {
        uint32_t j;
        const int gi = csv_locate(t, rows[k], j);
        const CsvGroup& g = t.g[gi];
        const int p = g.pass[j];
        if (p >= t.n_pass) continue;
        uint8_t* o = p >= 0 ? (out_m ? out_m + off_m[k] : nullptr) : (out_u ? out_u + off_u[k] : nullptr);
        if (!o) continue;
        const int L = csv_len(g, j);
        for (int w = 0; w * 32 < L; w++) {
            uint64_t bits = g.seq[(size_t)w * g.n + j];
            uint64_t nm = g.nmask ? g.nmask[(size_t)w * g.n + j] : 0ull;
            const int m = L - 32 * w < 32 ? L - 32 * w : 32;
            for (int b = 0; b < m; b++) {
                *o++ = (nm & 1ull) ? (uint8_t)'N' : (uint8_t)"ACGT"[bits & 3ull];
                bits >>= 2; nm >>= 2;
            }
        }
        *o++ = ',';
        *o++ = p >= 0 ? '1' : '0';
        const int col = p >= 0 ? t.col_of_pass[p] : -1;
        for (int c = 0; c < t.n_name_cols; c++) {
            *o++ = ',';
            if (c == col) {
                const int32_t r = g.ref[j];
                const uint32_t a = t.name_off[p][r], e = t.name_off[p][r + 1];
                const uint8_t* nd = t.name_data[p];
                for (uint32_t x = a; x < e; x++) *o++ = nd[x];
            }
        }
        for (int s = 0; s < t.S; s++) {
            *o++ = ',';
            uint32_t v = g.counts[(size_t)j * t.S + s];
            const int nd = csv_digits(v);
            for (int d = nd - 1; d >= 0; d--) { o[d] = (uint8_t)('0' + v % 10u); v /= 10u; }
            o += nd;
        }
        *o++ = '\n';
    }
}


// ------------------------------------------------------------------------------------------
// Row order of the sample matrix of SEVERAL samples: pandas' outer join sorts the union of the sequences
// (digest.py:243), i.e. Python string order -- A < C < G < N < T, a prefix before its extensions.  A read becomes a
// string of 3-bit digits (end 0, A 1, C 2, G 3, N 4, T 5), 21 to a 64-bit word, most significant first: comparing the
// words in turn IS that order.  k_lexkey writes word w of the reads listed in perm; the host runs a least-significant-word-
// first radix sort (stable pair sorts) over as many words as the longest read present needs.  Replaces a host-side
// argsort of a U x 32-byte matrix (3 s per 3 M unique reads; the whole GPU part of such a run is 0.3 s).
// ------------------------------------------------------------------------------------------
#define MIRGE_LEX_BASES 21
__global__ void k_lexkey(CsvTables t, const uint32_t* __restrict__ perm, uint32_t n, int32_t word, unsigned long long* __restrict__ keys) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        uint32_t j;
        const int gi = csv_locate(t, perm[k], j);
        const CsvGroup& g = t.g[gi];
        const int L = csv_len(g, j);
        unsigned long long key = 0ull;
        const int p0 = word * MIRGE_LEX_BASES;
#pragma unroll 1
        for (int b = 0; b < MIRGE_LEX_BASES; b++) {
            const int p = p0 + b;
            unsigned long long d = 0ull;
            if (p < L) {
                const uint64_t wd = g.seq[(size_t)(p >> 5) * g.n + j];
                const uint64_t nm = g.nmask ? g.nmask[(size_t)(p >> 5) * g.n + j] : 0ull;
                const uint32_t code = (uint32_t)((wd >> (2 * (p & 31))) & 3ull);
                const bool isn = (nm >> (2 * (p & 31))) & 1ull;
                d = isn ? 4ull : (code == 3u ? 5ull : (unsigned long long)code + 1ull);
            }
            key = (key << 3) | d;
        }
        keys[k] = key;
    }
}
// ------------------------------------------------------------------------------------------
// Range partition of a dictionary by the FIRST word of that order (round 6: the sharded run's parallel tail, multigpu.py).
// Rank q of a sharded run owns the reads whose word-0 key lies in [splitter[q-1], splitter[q]): reads that agree in their first
// 21 bases have one owner, and the owners' ranges are consecutive stretches of the run's sorted union (digest.py:243) -- so
// every rank can merge, annotate, order and format ITS stretch of mapped.csv / unmapped.csv (mirge/__main__.py:164-173).
//   k_range_owner : owner[i] = number of splitters <= key[i]                       (n_split <= 255)
//   k_invert_perm : pos[perm[k]] = k
//   k_part_bounds : bounds[q] = first position of part q in the owner-sorted order (bounds pre-filled with n)
//   k_scatter_rows: out[pos[base + j]][0..S) = counts[j][0..S)
// ------------------------------------------------------------------------------------------
__global__ void k_range_owner(const unsigned long long* __restrict__ keys, uint32_t n, const unsigned long long* __restrict__ split,
                              int32_t n_split, uint32_t* __restrict__ owner) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned long long k = keys[i];
        int lo = 0, hi = n_split;  // first splitter > k
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (split[mid] <= k) lo = mid + 1; else hi = mid; }
        owner[i] = (uint32_t)lo;
    }
}
__global__ void k_invert_perm(const uint32_t* __restrict__ perm, uint32_t n, uint32_t* __restrict__ pos) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) pos[perm[k]] = k;
}
__global__ void k_part_bounds(const uint32_t* __restrict__ owner_sorted, uint32_t n, unsigned long long* __restrict__ bounds) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const uint32_t cur = owner_sorted[k];
        const uint32_t from = k ? owner_sorted[k - 1] + 1u : 0u;
        for (uint32_t q = from; q <= cur; q++) bounds[q] = k;  // (parts nobody falls into start where the next one does)
    }
}
__global__ void k_scatter_rows(const uint32_t* __restrict__ counts, uint32_t n, int32_t S, const uint32_t* __restrict__ pos,
                               uint32_t* __restrict__ out) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
        for (int32_t s = 0; s < S; s++) out[(size_t)pos[j] * S + s] = counts[(size_t)j * S + s];
}
__global__ void k_iota(uint32_t* __restrict__ out, uint32_t n) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) out[k] = k;
}
// nz[s] = unique reads with a count in sample s ('Trimmed Reads (unique)' per sample, digest.py:214)
__global__ void k_count_nonzero_cols(const uint32_t* __restrict__ counts, uint32_t n, int32_t S, unsigned long long* __restrict__ nz) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        for (int32_t s = 0; s < S; s++) {
            const bool has = counts[(size_t)i * S + s] != 0u;
            const unsigned long long bal = __ballot(has);
            if (bal && (threadIdx.x & 63) == (unsigned)(__ffsll(bal) - 1)) atomicAdd(&nz[s], (unsigned long long)__popcll(bal));
        }
}

// ------------------------------------------------------------------------------------------
// The miRTop GFF3 (`-gff`, sample_miRge3.gff; create_gff, mirge/libs/summary.py:48-606) formatted on the device (round 6).  Until
// round 5 k_isotype's 336-byte records crossed PCIe (241 MB per 0.72 M miRNA reads) and mirge_gff_write built the 160 MB of text on
// host cores from them plus a host copy of every read, count and annotation; the file was 0.12 of a sample's 0.28 s.  Now the
// rows are chosen, typed, measured and written where the reads, counts, annotation and records already lie:
//   k_gff_select : frame position k -> flags "exact-miRNA row" / "isomiR row" (two exclusive scans give the file's row order:
//                  the exact rows of the mapped frame, then its isomiR rows, summary.py:50-60)
//   k_gff_rows   : row -> the read's handle index and its slot for k_isotype
//   k_gff_line<false> : bytes of every line;  k_gff_line<true> : the lines, each at its offset
// A line:  NAME \t SOURCE \t ref_miRNA|isomiR \t START \t END \t.\t+\t.\tRead=SEQ; UID=...; Name=NAME; Parent=PRE; Variant=V; Cigar=C;
//          Expression=c0,c1,..; Filter=Pass; Hits=c0,c1,..\n      -- mirge_gff_write's bytes (native_host.hpp), which the golden files pin.
// ------------------------------------------------------------------------------------------
struct GffTables {
    const int32_t* name_of_ref;     // [n_mirna] printed-name index of a miRNA reference, -1: none
    const int32_t* parent_of_ref;   // [n_mirna]
    const uint8_t* name_data;       // printed names back to back
    const uint32_t* name_off;       // [n_names + 1]
    const uint8_t* parent_data;
    const uint32_t* parent_off;     // [n_parents + 1]
    uint32_t n_names, n_parents, n_mirna;
    const uint8_t* source;          // e.g. "miRBase22"
    uint32_t source_len;
};

__global__ void k_gff_select(CsvTables t, const uint32_t* __restrict__ order, uint32_t n, int32_t exact_pass, int32_t iso_pass,
                             uint32_t* __restrict__ f_exact, uint32_t* __restrict__ f_iso) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        uint32_t j;
        const int gi = csv_locate(t, order[k], j);
        const int p = t.g[gi].pass[j];
        f_exact[k] = (p == exact_pass && p >= 0) ? 1u : 0u;
        f_iso[k] = (p == iso_pass && p >= 0) ? 1u : 0u;
    }
}
__global__ void k_gff_rows(const uint32_t* __restrict__ order, uint32_t n, const uint32_t* __restrict__ f_exact, const uint32_t* __restrict__ f_iso,
                           const uint32_t* __restrict__ p_exact, const uint32_t* __restrict__ p_iso, uint32_t n_exact,
                           uint32_t* __restrict__ rows, int32_t* __restrict__ slot_of_read) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const uint32_t read = order[k];
        if (f_exact[k]) { rows[p_exact[k]] = read; slot_of_read[read] = (int32_t)p_exact[k]; }
        else if (f_iso[k]) { rows[n_exact + p_iso[k]] = read; slot_of_read[read] = (int32_t)(n_exact + p_iso[k]); }
    }
}

// a cursor that counts (WRITE = false) or writes
template <bool WRITE>
struct GffOut {
    uint8_t* o;
    uint32_t n;
    __device__ __forceinline__ void ch(char c) { if (WRITE) o[n] = (uint8_t)c; n++; }
    __device__ __forceinline__ void str(const char* s) { for (int k = 0; s[k]; k++) ch(s[k]); }
    __device__ __forceinline__ void bytes(const uint8_t* s, uint32_t len) { if (WRITE) for (uint32_t k = 0; k < len; k++) o[n + k] = s[k]; n += len; }
    __device__ __forceinline__ void uint(uint32_t v) {
        const int nd = csv_digits(v);
        if (WRITE) { uint32_t x = v; for (int d = nd - 1; d >= 0; d--) { o[n + d] = (uint8_t)('0' + x % 10u); x /= 10u; } }
        n += (uint32_t)nd;
    }
    __device__ __forceinline__ void sint(int32_t v) { if (v < 0) { ch('-'); uint((uint32_t)(-(int64_t)v)); } else uint((uint32_t)v); }
};

// flags[0] |= 1: a name index out of range or a record's text beyond its capacity
template <bool WRITE>
__global__ void k_gff_line(CsvTables t, GffTables gt, const uint32_t* __restrict__ rows, uint32_t n_rows, const MirgeIsoRec* __restrict__ recs,
                           unsigned long long* __restrict__ len_out, const unsigned long long* __restrict__ off, uint8_t* __restrict__ out,
                           uint32_t* __restrict__ flags) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_rows; k += gridDim.x * blockDim.x) {
        const MirgeIsoRec& r = recs[k];
        if (r.kind == 0) { if (!WRITE) len_out[k] = 0ull; continue; }  // a name the reference has no annotation for: no line (summary.py:492)
        uint32_t j;
        const int gi = csv_locate(t, rows[k], j);
        const CsvGroup& g = t.g[gi];
        const int32_t ref = g.ref[j];
        const int32_t ni = (ref >= 0 && (uint32_t)ref < gt.n_mirna) ? gt.name_of_ref[ref] : -1;
        const int32_t pi = (ref >= 0 && (uint32_t)ref < gt.n_mirna) ? gt.parent_of_ref[ref] : -1;
        if (ni < 0 || (uint32_t)ni >= gt.n_names || pi < 0 || (uint32_t)pi >= gt.n_parents || (uint32_t)r.vlen + r.clen > MIRGE_ISO_TEXT) {
            atomicOr(&flags[0], 1u);
            if (!WRITE) len_out[k] = 0ull;
            continue;
        }
        GffOut<WRITE> w{WRITE ? out + off[k] : nullptr, 0u};
        const uint8_t* nm = gt.name_data + gt.name_off[ni];
        const uint32_t nl = gt.name_off[ni + 1] - gt.name_off[ni];
        w.bytes(nm, nl); w.ch('\t'); w.bytes(gt.source, gt.source_len); w.ch('\t');
        w.str(r.kind == 1 ? "ref_miRNA" : "isomiR");
        w.ch('\t'); w.sint(r.start); w.ch('\t'); w.sint(r.end);
        w.str("\t.\t+\t.\tRead=");
        const int L = csv_len(g, j);
        bool has_n = false;
        for (int wd = 0; wd * 32 < L; wd++) {
            uint64_t bits = g.seq[(size_t)wd * g.n + j];
            uint64_t nmk = g.nmask ? g.nmask[(size_t)wd * g.n + j] : 0ull;
            const int m = L - 32 * wd < 32 ? L - 32 * wd : 32;
            for (int b = 0; b < m; b++) {
                const bool isn = nmk & 1ull;
                has_n |= isn;
                w.ch(isn ? 'N' : "ACGT"[bits & 3ull]);
                bits >>= 2; nmk >>= 2;
            }
        }
        w.str("; UID=");
        if (has_n) w.ch('.');
        else {  // miRgeEssential.UID (:364-370): see uid_append (native_host.hpp)
            w.str(r.kind == 1 ? "ref-" : "iso-"); w.uint((uint32_t)L); w.ch('-');
            for (int at = 0; at < L; at += 5) {
                const int kk = L - at < 5 ? L - at : 5;
                int v = 0;
                for (int q = 0; q < kk; q++) {
                    const int p = at + q;
                    v = v * 4 + (int)((g.seq[(size_t)(p >> 5) * g.n + j] >> (2 * (p & 31))) & 3ull);
                }
                if (kk == 5) { w.ch("BD0EF1HI2JK3LM4NO5PQ6RS7UV8WX9YZ"[v / 32]); w.ch("BD0EF1HI2JK3LM4NO5PQ6RS7UV8WX9YZ"[v % 32]); }
                else {
                    v += kk == 1 ? 0 : kk == 2 ? 4 : kk == 3 ? 20 : 84;
                    if (v < 32) w.ch("BD0EF1HI2JK3LM4NO5PQ6RS7UV8WX9YZ"[v]);
                    else { w.ch("BD0EF1HI2JK3LM4NO5PQ6RS7UV8WX9YZ"[v / 32]); w.ch("BD0EF1HI2JK3LM4NO5PQ6RS7UV8WX9YZ"[v % 32]); }
                }
            }
        }
        w.str("; Name="); w.bytes(nm, nl);
        w.str("; Parent="); w.bytes(gt.parent_data + gt.parent_off[pi], gt.parent_off[pi + 1] - gt.parent_off[pi]);
        w.str("; Variant="); w.bytes(reinterpret_cast<const uint8_t*>(r.text), r.vlen);
        w.str("; Cigar="); w.bytes(reinterpret_cast<const uint8_t*>(r.text) + r.vlen, r.clen);
        w.str("; Expression=");
        for (int s = 0; s < t.S; s++) { if (s) w.ch(','); w.uint(g.counts[(size_t)j * t.S + s]); }
        w.str("; Filter=Pass; Hits=");
        for (int s = 0; s < t.S; s++) { if (s) w.ch(','); w.uint(g.counts[(size_t)j * t.S + s]); }
        w.ch('\n');
        if (!WRITE) len_out[k] = (unsigned long long)w.n;
    }
}
