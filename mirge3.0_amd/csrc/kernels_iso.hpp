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

template <int W>
__global__ void k_isotype(GroupView<W> g, uint32_t base, const uint32_t* __restrict__ orig,
                          const int32_t* __restrict__ res_ref, IsoTables tb,
                          const int32_t* __restrict__ slot_of_read, MirgeIsoRec* __restrict__ out,
                          const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_list, uint32_t chunk) {
    // the workgroup's chunk of reads, compacted to its miRNA rows (k_member_list<IsoMember>: kernels_join.hpp)
    const uint32_t n_rows = n_list[blockIdx.x];
    for (uint32_t k = threadIdx.x; k < n_rows; k += blockDim.x) {
        const uint32_t i = list[(size_t)blockIdx.x * chunk + k];
        const uint32_t h = orig ? orig[i] : base + i;
        const int32_t slot = slot_of_read[h];
        MirgeIsoRec rec;
        rec.kind = 0; rec.reserved = 0; rec.start = rec.end = 0; rec.vlen = rec.clen = 0;
        const int32_t mi = tb.master_of_ref[res_ref[i]];
        MirgeRead<W> r;
        load_read<W>(g, i, r);
        if (mi >= 0 && r.len <= MIRGE_ISO_MAXB) {
            char b[MIRGE_ISO_MAXB + 1], a[MIRGE_ISO_MAXA + 1];
            for (int k = 0; k < r.len; k++) {
                const bool n = (r.nm[k >> 5] >> (2 * (k & 31))) & 1ull;
                b[k] = n ? 'N' : "ACGT"[(r.w[k >> 5] >> (2 * (k & 31))) & 3ull];
            }
            const int32_t a0 = tb.master_off[mi], la = tb.master_off[mi + 1] - a0;
            if (la <= MIRGE_ISO_MAXA) {
                for (int k = 0; k < la; k++) a[k] = tb.master[a0 + k];
                const int32_t pi = tb.pre_of_master[mi];
                const int32_t p0 = tb.pre_off[pi], lp = tb.pre_off[pi + 1] - p0;
                mirge_isotype(a, la, b, r.len, tb.pre + p0, lp, tb.start0[mi], rec);
            }
        }
        MirgeIsoRec* dst = out + slot;
        dst->start = rec.start; dst->end = rec.end; dst->kind = rec.kind; dst->reserved = 0;
        dst->vlen = rec.vlen; dst->clen = rec.clen;
        const int nt = rec.vlen + rec.clen;
        for (int k = 0; k < nt; k++) dst->text[k] = rec.text[k];
    }
}
