// kernels_iso.hpp -- part of mirge_kernels.hpp: k_isotype, the per-read isomiR typing of the miRTop GFF3 (row N2).
#pragma once
#include "mirge_isotype.hpp"

// What the reference looks up per miRNA name before it types a read (summary.py:147-186): the canonical sequence of
// the name (mature FASTA), its precursor (annotation GFF3 + hairpin index) and where the canonical sits in it.
struct IsoTables {
    const int32_t* master_of_ref;  // [n_mirna] row of the tables below for every reference of the miRNA library, -1: none
    const char* master;            // canonical sequences, ASCII
    const int32_t* master_off;     // [n_master + 1]
    const int32_t* pre_of_master;  // [n_master] precursor of each canonical
    const int32_t* start0;         // [n_master] precursor.find(canonical) + 1 (1 when the precursor is empty)
    const char* pre;               // precursor sequences, ASCII
    const int32_t* pre_off;        // [n_pre + 1]
};

// the read's letters as code bit planes, straight from its packed words (2 bits per base at bit 2p, N mask at bit 2p)
__device__ __forceinline__ uint64_t iso_even_bits(uint64_t x) {  // bits 0, 2, 4, ... -> bits 0, 1, 2, ...
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    return x;
}
template <int W>
__device__ __forceinline__ void iso_seq_of_read(const MirgeRead<W>& r, mirge_iso::Seq& q) {
    static_assert(W <= 2, "k_isotype types reads of up to 64 nt");
    uint64_t c0 = iso_even_bits(r.w[0]), c1 = iso_even_bits(r.w[0] >> 1), n = iso_even_bits(r.nm[0]);
    if (W == 2) {
        c0 |= iso_even_bits(r.w[W - 1]) << 32; c1 |= iso_even_bits(r.w[W - 1] >> 1) << 32; n |= iso_even_bits(r.nm[W - 1]) << 32;
    }
    const uint64_t v = mirge_iso::lowmask(r.len);
    n &= v;
    q.c0 = c0 & v & ~n; q.c1 = c1 & v & ~n; q.c2 = n; q.n = r.len;
}

struct IsoLdsWs {  // the matching blocks and the recursion's work list of one thread: column threadIdx.x of a [2 x 16][64] LDS array
    uint32_t* base;
    __device__ __forceinline__ uint32_t& blk(int k) { return base[k * 64]; }
    __device__ __forceinline__ uint32_t& que(int k) { return base[(MIRGE_ISO_FAST_BLOCKS + k) * 64]; }
};

#ifndef MIRGE_ISO_FAST
#define MIRGE_ISO_FAST 1  // 0: every read through the array form (A/B, and the tests' second implementation on the device)
#endif

#ifndef MIRGE_ISO_WAVES
#define MIRGE_ISO_WAVES 4  // waves per SIMD the register allocation has to leave room for (128 VGPRs)
#endif
template <int W>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MIRGE_ISO_WAVES, 8))) k_isotype(GroupView<W> g, uint32_t base, const uint32_t* __restrict__ orig,
                          const int32_t* __restrict__ res_ref, IsoTables tb,
                          const int32_t* __restrict__ slot_of_read, MirgeIsoRec* __restrict__ out,
                          const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_list, uint32_t chunk, int32_t fast) {
    __shared__ uint32_t s_ws[2 * MIRGE_ISO_FAST_BLOCKS][64];
    // the workgroup's chunk of reads, compacted to its miRNA rows (k_member_list<IsoMember>: kernels_join.hpp)
    const uint32_t n_rows = n_list[blockIdx.x];
    for (uint32_t k = threadIdx.x; k < n_rows; k += blockDim.x) {
        const uint32_t i = list[(size_t)blockIdx.x * chunk + k];
        const uint32_t h = orig ? orig[i] : base + i;
        const int32_t slot = slot_of_read[h];
        MirgeIsoRec* dst = out + slot;
        const int32_t mi = tb.master_of_ref[res_ref[i]];
        MirgeRead<W> r;
        load_read<W>(g, i, r);
        const bool typed = mi >= 0 && r.len <= MIRGE_ISO_MAXB;
        int32_t a0 = 0, la = 0, pi = 0, p0 = 0, lp = 0;
        if (typed) {
            a0 = tb.master_off[mi]; la = tb.master_off[mi + 1] - a0;
            pi = tb.pre_of_master[mi];
            p0 = tb.pre_off[pi]; lp = tb.pre_off[pi + 1] - p0;
        }
        if (MIRGE_ISO_FAST && fast && typed && la <= MIRGE_ISO_MAXA) {  // everything in registers (mirge_isotype_fast)
            mirge_iso::Seq qa, qb;
            iso_seq_of_read<W>(r, qb);
            if (mirge_iso::seq_of_ascii(tb.master + a0, la, qa)) {
                IsoLdsWs ws;
                ws.base = &s_ws[0][threadIdx.x];
                dst->reserved = 0;
                if (mirge_isotype_fast(qa, qb, tb.pre + p0, lp, tb.start0[mi], ws, &dst->start, &dst->end, &dst->kind, &dst->vlen, &dst->clen, dst->text))
                    continue;
            }
        }
        MirgeIsoRec rec;
        rec.kind = 0; rec.reserved = 0; rec.start = rec.end = 0; rec.vlen = rec.clen = 0;
        if (typed) {
            char b[MIRGE_ISO_MAXB + 1], a[MIRGE_ISO_MAXA + 1];
            for (int k = 0; k < r.len; k++) {
                const bool n = (r.nm[k >> 5] >> (2 * (k & 31))) & 1ull;
                b[k] = n ? 'N' : "ACGT"[(r.w[k >> 5] >> (2 * (k & 31))) & 3ull];
            }
            if (la <= MIRGE_ISO_MAXA) {
                for (int k = 0; k < la; k++) a[k] = tb.master[a0 + k];
                mirge_isotype(a, la, b, r.len, tb.pre + p0, lp, tb.start0[mi], rec);
            }
        }
        dst->start = rec.start; dst->end = rec.end; dst->kind = rec.kind; dst->reserved = 0;
        dst->vlen = rec.vlen; dst->clen = rec.clen;
        const int nt = rec.vlen + rec.clen;
        for (int k = 0; k < nt; k++) dst->text[k] = rec.text[k];
    }
}
