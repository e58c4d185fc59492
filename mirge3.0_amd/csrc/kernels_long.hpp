// kernels_long.hpp -- part of mirge_kernels.hpp: the LONG read class (reads of more than 255 nt).
//
// The reference has no upper bound on a read's length: `-M` is parsed and never read (mirge/libs/parse.py:102), the worker
// tests the minimum only (digest.py:348,368), and an untrimmed 300-cycle read whose adapter was not found simply goes through
// bowtie.  Such reads are rare -- none in a trimmed small-RNA sample -- so they get a class of their own whose kernels are
// written for any length and no speed: W 64-bit words per read with W fixed per read SET (the longest read present), the
// length in 16 bits (up to 65535 nt), one thread per read, words fetched from memory as they are needed.  Same layout
// (word-major, 2 bits per base + an N mask), same answers: the alignment is the arithmetic of mirge_window_mm /
// mirge_window_invalid word by word, its candidates come from the probe plan of the read's first 31 bases (any alignment
// within the mismatch budget stays within it on a prefix), ranked as everywhere: (member library, mismatches, position).
#pragma once

struct LongView {
    const uint64_t* seq;    // [W][n]
    const uint64_t* nmask;  // [W][n] or nullptr
    const uint16_t* len;    // [n]
    uint32_t n;
    uint32_t W;
};

__global__ void k_pack_long(const uint8_t* __restrict__ ascii, const int64_t* __restrict__ starts, const int64_t* __restrict__ ends,
                            const uint32_t* __restrict__ idx, uint32_t n, uint32_t W, uint64_t* __restrict__ seq,
                            uint16_t* __restrict__ len, uint64_t* __restrict__ nmask, uint32_t* __restrict__ flags,
                            const int64_t* __restrict__ s2start, const int32_t* __restrict__ s2len) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const uint32_t src = idx[j];
        const int64_t b = starts[src];
        const int L1 = (int)(ends[src] - b);
        const int L2 = s2len ? (int)s2len[src] : 0;
        const int64_t b2 = s2len ? s2start[src] - L1 : 0;
        const int L = L1 + L2;
        uint32_t sawN = 0, bad = 0, iupac = 0;
        uint64_t w = 0, nm = 0;
        uint32_t wi = 0;
        for (int p = 0; p < L; p++) {
            const uint8_t c = (p < L1 ? ascii[b + p] : ascii[b2 + p]) & 0xDF;
            const uint32_t bit = letter_bit(c);
            const uint32_t x = (c >> 1) & 3u;
            const uint64_t isn = (bit & MIRGE_LETTERS_ACGTU) ? 0ull : 1ull;
            const uint64_t code = isn ? 0ull : (uint64_t)(x ^ (x >> 1));
            iupac |= (bit & MIRGE_LETTERS_IUPAC) != 0;
            bad |= isn && !(bit & (MIRGE_LETTERS_N | MIRGE_LETTERS_IUPAC));
            sawN |= (uint32_t)isn;
            w |= code << (2 * (p & 31));
            nm |= isn << (2 * (p & 31));
            if ((p & 31) == 31 || p == L - 1) {
                seq[(size_t)wi * n + j] = w;
                nmask[(size_t)wi * n + j] = nm;
                w = 0; nm = 0; wi++;
            }
        }
        for (; wi < W; wi++) { seq[(size_t)wi * n + j] = 0ull; nmask[(size_t)wi * n + j] = 0ull; }
        len[j] = (uint16_t)L;
        if (sawN) atomicOr(&flags[0], 1u | (iupac << 1));
        if (bad) atomicOr(&flags[1], 1u);
    }
}

__global__ void k_unpack_long(LongView g, const int64_t* __restrict__ out_off, uint32_t base, const uint32_t* __restrict__ orig,
                              uint8_t* __restrict__ ascii_out) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < g.n; j += gridDim.x * blockDim.x) {
        const uint32_t dst = orig ? orig[j] : base + j;
        uint8_t* o = ascii_out + out_off[dst];
        const int L = g.len[j];
        for (int p = 0; p < L; p++) {
            const uint64_t w = g.seq[(size_t)(p >> 5) * g.n + j];
            const uint64_t m = g.nmask ? g.nmask[(size_t)(p >> 5) * g.n + j] : 0ull;
            o[p] = ((m >> (2 * (p & 31))) & 1ull) ? 'N' : "ACGT"[(w >> (2 * (p & 31))) & 3ull];
        }
    }
}

__global__ void k_scatter_len16(const uint16_t* __restrict__ len, uint32_t n, uint32_t base, const uint32_t* __restrict__ orig,
                                int32_t* __restrict__ out) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
        out[orig ? orig[j] : base + j] = len[j];
}

// words [0, W) of one read appended behind `at` reads of a wider or equally wide set (mirge_reads_concat: the parts may have
// been packed for different longest reads)
__global__ void k_long_copy(LongView src, uint32_t dstW, uint32_t dst_n, uint32_t at, uint64_t* __restrict__ dseq,
                            uint64_t* __restrict__ dnmask, uint16_t* __restrict__ dlen) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < src.n; j += gridDim.x * blockDim.x) {
        for (uint32_t w = 0; w < dstW; w++) {
            dseq[(size_t)w * dst_n + at + j] = w < src.W ? src.seq[(size_t)w * src.n + j] : 0ull;
            if (dnmask) dnmask[(size_t)w * dst_n + at + j] = (w < src.W && src.nmask) ? src.nmask[(size_t)w * src.n + j] : 0ull;
        }
        dlen[at + j] = src.len[j];
    }
}

// ---- collapse (digest.py:141-163), the general path's three steps for reads of any length -------------------------------
__device__ __forceinline__ bool long_same(const LongView& g, uint32_t a, uint32_t b) {
    if (g.len[a] != g.len[b]) return false;
    const uint32_t nw = ((uint32_t)g.len[a] + 31u) >> 5;
    for (uint32_t w = 0; w < nw; w++) {
        if (g.seq[(size_t)w * g.n + a] != g.seq[(size_t)w * g.n + b]) return false;
        if (g.nmask && g.nmask[(size_t)w * g.n + a] != g.nmask[(size_t)w * g.n + b]) return false;
    }
    return true;
}

__global__ void k_collapse_insert_long(LongView g, uint32_t* __restrict__ rep, uint32_t* __restrict__ firstj, uint32_t* __restrict__ cnt,
                                       uint32_t* __restrict__ slot_of, uint32_t mask, const int32_t* __restrict__ sample_ids,
                                       const uint32_t* __restrict__ orig, uint32_t base, int32_t S, const uint32_t* __restrict__ weight) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < g.n; j += gridDim.x * blockDim.x) {
        const uint32_t nw = ((uint32_t)g.len[j] + 31u) >> 5;
        uint64_t h = mirge_mix64((uint64_t)g.len[j] << 40);
        for (uint32_t w = 0; w < nw; w++) {
            h = mirge_mix64(h ^ g.seq[(size_t)w * g.n + j]);
            if (g.nmask) h ^= mirge_mix64(g.nmask[(size_t)w * g.n + j] + 0x9e3779b97f4a7c15ull * (w + 1));
        }
        uint32_t s = (uint32_t)(h >> 20) & mask;
        while (true) {
            uint32_t cur = rep[s];
            if (cur == MIRGE_EMPTY) cur = atomicCAS(&rep[s], MIRGE_EMPTY, j);
            if (cur == MIRGE_EMPTY || cur == j || long_same(g, cur, j)) break;
            s = (s + 1) & mask;
        }
        slot_of[j] = s;
        const uint32_t hidx = orig ? orig[j] : base + j;
        const int32_t sid = sample_ids ? sample_ids[hidx] : 0;
        atomicMin(&firstj[s], j);
        atomicAdd(&cnt[(size_t)s * S + sid], weight ? weight[hidx] : 1u);
    }
}

// the heads' total length (mirge_reads_total_bases of the collapsed set: the length histogram stops at 255)
__global__ void k_long_heads_bases(const uint32_t* __restrict__ slot_of, const uint32_t* __restrict__ firstj, const uint16_t* __restrict__ len,
                                   uint32_t n, unsigned long long* __restrict__ total) {
    unsigned long long t = 0;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
        if (firstj[slot_of[j]] == j) t += len[j];
    if (t) atomicAdd(total, t);
}

__global__ void k_collapse_scatter_long(LongView g, const uint32_t* __restrict__ slot_of, const uint8_t* __restrict__ flag,
                                        const uint32_t* __restrict__ cnt, uint32_t cnt_stride, const uint32_t* __restrict__ blockoff,
                                        const uint32_t* __restrict__ n_uniq_ptr, const uint32_t* __restrict__ orig, uint32_t base, int32_t S,
                                        uint64_t* __restrict__ useq, uint16_t* __restrict__ ulen, uint64_t* __restrict__ unmask,
                                        uint32_t* __restrict__ ucnt, uint32_t* __restrict__ ufirst) {
    __shared__ uint32_t lds4[4];
    const uint32_t U = *n_uniq_ptr;
    const uint32_t b0 = blockIdx.x * (MIRGE_BLOCK * MIRGE_SCAN_ITEMS) + threadIdx.x * MIRGE_SCAN_ITEMS;
    uint32_t heads = 0, c = 0;
#pragma unroll
    for (int i = 0; i < MIRGE_SCAN_ITEMS; i++) if (b0 + i < g.n && flag[b0 + i]) { heads |= 1u << i; c++; }
    uint32_t total;
    uint32_t rank = blockoff[blockIdx.x] + block_excl_scan(c, total, lds4);
#pragma unroll
    for (int i = 0; i < MIRGE_SCAN_ITEMS; i++) {
        if (heads & (1u << i)) {
            const uint32_t j = b0 + i;
            const uint32_t s = slot_of[j];
            for (uint32_t w = 0; w < g.W; w++) {
                useq[(size_t)w * U + rank] = g.seq[(size_t)w * g.n + j];
                if (unmask) unmask[(size_t)w * U + rank] = g.nmask ? g.nmask[(size_t)w * g.n + j] : 0ull;
            }
            ulen[rank] = g.len[j];
            for (int32_t q = 0; q < S; q++) ucnt[(size_t)rank * S + q] = cnt[(size_t)s * cnt_stride + q];
            ufirst[rank] = orig ? orig[j] : base + j;
            rank++;
        }
    }
}

// ---- cascade ------------------------------------------------------------------------------------------------------------
// the effective read of a pass as a VIEW of the stored one: bases [t5, t5 + l) (mirge_effective_read without moving anything)
struct LongRead {
    const uint64_t* seq;
    const uint64_t* nmask;
    size_t stride;  // n of the set: word w of the read is seq[w * stride]
    int t5, l;
};
__device__ __forceinline__ uint64_t long_bits(const uint64_t* plane, size_t stride, int nw_total, int base0, int k) {
    // bases [base0, base0 + k) (k <= 32) of a word-major sequence as an integer; words beyond the stored ones are zero
    const int q = base0 >> 5, s = (base0 & 31) * 2;
    uint64_t lo = q < nw_total ? plane[(size_t)q * stride] >> s : 0ull;
    if (s && q + 1 < nw_total) lo |= plane[(size_t)(q + 1) * stride] << (64 - s);
    return lo & mirge_lowmask2(k);
}

// mirge_window_mm + mirge_window_invalid for a read of any length at global position g; -1: outside the policy's budget
__device__ __forceinline__ int long_window_mm(const MirgeLibView& lib, const MirgePolicy& p, const LongRead& r, int nw_total, uint64_t g) {
    const int L = r.l;
    if (g + (uint64_t)L > lib.total) return -1;
    const int seed = p.mode == 0 ? (L < p.seedlen ? L : p.seedlen) : L;
    int tot = 0, seedmm = 0;
    for (int i = 0; 32 * i < L; i++) {
        const int rem = L - 32 * i, k = rem > 32 ? 32 : rem;
        const uint64_t gp = g + 32ull * i;
        const uint64_t q = gp >> 5;
        const int s = (int)(gp & 31) * 2;
        uint64_t t = lib.T[q] >> s;
        if (s) t |= lib.T[q + 1] << (64 - s);
        const uint64_t x = long_bits(r.seq, r.stride, nw_total, r.t5 + 32 * i, k) ^ (t & mirge_lowmask2(k));
        uint64_t m = (x | (x >> 1)) & 0x5555555555555555ull & mirge_lowmask2(k);
        if (r.nmask) m |= long_bits(r.nmask, r.stride, nw_total, r.t5 + 32 * i, k);
        tot += mirge_popc(m);
        const int srem = seed - 32 * i;
        if (srem > 0) seedmm += mirge_popc(m & mirge_lowmask2(srem > 32 ? 32 : srem));
        if (tot > p.maxtotal || seedmm > p.mm) return -1;
    }
    if (mirge_window_invalid(lib.inv, g, L)) return -1;
    return tot;
}

// all steps of the cascade for the reads of a long group, one thread per read (k_cascade_fused's structure: no compaction)
__global__ void k_cascade_long(const FusedSteps* __restrict__ steps, ResolveTable tb, LongView g, int8_t* __restrict__ res_pass,
                               uint32_t* __restrict__ res_pos, int8_t* __restrict__ res_mm, int32_t* __restrict__ res_ref,
                               int32_t* __restrict__ res_off) {
    const int nsteps = steps->n;
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < g.n; idx += gridDim.x * blockDim.x) {
        const int L0 = g.len[idx];
        const int nw_total = (int)g.W;
        int8_t o_pass = -1, o_mm = -1;
        uint32_t o_pos = 0;
        for (int si = 0; si < nsteps && o_pass < 0; si++) {
            const FusedStep& st = steps->s[si];
            const MirgePolicy& p = st.pol;
            // what bowtie is handed (mirge_effective_read)
            if (p.len_lt > 0 && !(L0 < p.len_lt)) continue;
            if (p.len_gt > 0 && !(L0 > p.len_gt)) continue;
            int l = L0;
            if (p.ttail) {
                int run = 0;
                for (int j = L0 - 1; j >= 0; j--) {
                    const uint64_t b = (g.seq[(size_t)(j >> 5) * g.n + idx] >> (2 * (j & 31))) & 3ull;
                    const uint64_t nn = g.nmask ? (g.nmask[(size_t)(j >> 5) * g.n + idx] >> (2 * (j & 31))) & 1ull : 0ull;
                    if (b != 3ull || nn) break;
                    run++;
                }
                if (run < 3) continue;
                l = L0 - run;
            }
            l -= p.trim5 + p.trim3;
            if (l < 1 || l <= p.mm) continue;
            LongRead r{g.seq + idx, g.nmask ? g.nmask + idx : nullptr, (size_t)g.n, p.trim5, l};
            // candidates: the probe plan of the first min(l, 31) bases -- a one-word read in its own right
            MirgeRead<1> pre;
            const int pl = l < 31 ? l : 31;
            pre.w[0] = long_bits(r.seq, r.stride, nw_total, r.t5, pl);
            pre.nm[0] = r.nmask ? long_bits(r.nmask, r.stride, nw_total, r.t5, pl) : 0ull;
            pre.len = pl;
            const int np = st.plan->np[pl];
            uint64_t best = MIRGE_NO_HIT;
            for (int q = 0; q < np; q++) {
                const MirgeProbe pr = st.plan->pr[pl][q];
                uint64_t key;
                if (!mirge_probe_key<1>(pre, pr, key)) continue;
                const MirgeKTable t = st.lib.tables[mirge_shape_id(pr.k1, pr.gap, pr.k2)];
                uint32_t lo, cnt;
                bool inl = false;
                bool entries;
                gptr_u32 tbits = table_bits(t, entries);  // (MIRGE_PRESENCE_FILTER: the pointer may carry the "entries behind a filter" mark)
                if (tbits && !((tbits[key >> 5] >> (key & 31)) & 1u)) continue;
                if (!entries) {
                    const uint32_t* b = static_cast<const uint32_t*>(t.bucket);
                    lo = b[key]; cnt = b[key + 1] - lo;
                } else {
                    const uint64_t e = static_cast<const uint64_t*>(t.bucket)[key];
                    cnt = (uint32_t)(e >> 32); lo = (uint32_t)e; inl = cnt == 1;
                }
                for (uint32_t c = 0; c < cnt; c++) {
                    const uint32_t pz = inl ? lo : t.pos[lo + c];
                    if (pz < (uint32_t)pr.a1) continue;
                    const uint64_t gpos = (uint64_t)pz - pr.a1;
                    const int m = long_window_mm(st.lib, p, r, nw_total, gpos);
                    if (m < 0) continue;
                    const uint64_t cand = class_key(st.mi, gpos) | ((uint64_t)m << 32) | gpos;
                    if (cand < best) best = cand;
                }
            }
            if (best != MIRGE_NO_HIT) {
                const int cls = (int)(best >> 40);
                o_pass = (int8_t)(st.pass_id + cls);
                uint32_t b0 = 0;
                for (int i = 1; i < 4; i++) if (i == cls) b0 = st.mi.bound[i];
                o_pos = (uint32_t)best - b0;
                o_mm = (int8_t)((best >> 32) & 0xFF);
            }
        }
        int32_t ref = -1, off = -1;
        if (o_pass >= 0) resolve_one(tb, o_pass, o_pos, ref, off);
        res_pass[idx] = o_pass; res_pos[idx] = o_pos; res_mm[idx] = o_mm;
        res_ref[idx] = ref; res_off[idx] = off;
    }
}
