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
