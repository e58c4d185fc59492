// mirge_isotype.hpp -- isomiR typing of one read against its miRNA for the miRTop GFF3:
// what create_gff (mirge/libs/summary.py:170-470) derives per read with difflib.Differ and a series of in-place list
// rewrites, as one function on fixed-size arrays.  Compiled into k_isotype (kernels_join.hpp) and, for logic tests
// without a GPU, into tests/hostsim.  Everything is byte / index arithmetic.
//
//   1. difflib.SequenceMatcher(None, canonical, read): longest matching block first (ties: earliest in the
//      canonical, then earliest in the read), recursively left and right of it (get_matching_blocks); no junk,
//      autojunk never triggers below 200 elements.
//   2. Differ.compare: an 'equal' block prints its elements, 'delete' / 'insert' print '-' / '+' lines, 'replace'
//      (no common element inside, so _fancy_replace always falls through to _plain_replace) prints the SHORTER side
//      first, the canonical's '-' lines first when both have one length.
//   3. summary.py:226-289: two aligned lists -- the canonical with '-' at inserted columns, the read as '_' (deleted),
//      '+X' (inserted) or 'X' -- then a forward and a reverse pass that fold "deleted then inserted" pairs into one
//      substituted column.  The passes delete from the lists they iterate over; they are restated as index loops with
//      Python's list-iterator semantics, including the IndexError they swallow.
//   4. :297-422: additions / deletions at the two ends (templated against the precursor or not), substitutions
//      classed by column index, the variant string; :427-463: the CIGAR-like string.
#pragma once
#include <stdint.h>

#ifndef MIRGE_HD
#if defined(__HIPCC__)
#define MIRGE_HD __host__ __device__ __forceinline__
#else
#define MIRGE_HD inline
#endif
#endif

#define MIRGE_ISO_MAXA 40                                  // canonical miRNA, nt
#define MIRGE_ISO_MAXB 64                                  // read, nt
#define MIRGE_ISO_MAXN (MIRGE_ISO_MAXA + MIRGE_ISO_MAXB)   // aligned columns
#define MIRGE_ISO_TEXT 320                                 // variant + cigar bytes of one record

struct MirgeIsoRec {
    int32_t start, end;   // coordinates on the precursor, 1-based (summary.py:181-186 and the shifts of :356-402)
    uint8_t kind;         // 0: not typed (no annotation for the miRNA), 1: ref_miRNA, 2: isomiR
    uint8_t reserved;
    uint16_t vlen, clen;  // text = variant (vlen bytes) then cigar (clen bytes)
    char text[MIRGE_ISO_TEXT];
};

namespace mirge_iso {

struct Lists {  // the two aligned lists of step 3
    char m[MIRGE_ISO_MAXN + 2];   // canonical base or '-'
    char st[MIRGE_ISO_MAXN + 2];  // '_' deleted, '+' inserted, '=' equal
    char sc[MIRGE_ISO_MAXN + 2];  // the read's base for '+' and '='
    int n;
};

MIRGE_HD void put(MirgeIsoRec& o, int& at, const char* s) {
    for (int k = 0; s[k] && at < MIRGE_ISO_TEXT - 1; k++) o.text[at++] = s[k];
}
MIRGE_HD void put_int(MirgeIsoRec& o, int& at, int v) {
    char buf[12];
    int k = 0;
    if (v < 0) { if (at < MIRGE_ISO_TEXT - 1) o.text[at++] = '-'; v = -v; }
    do { buf[k++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (k > 0 && at < MIRGE_ISO_TEXT - 1) o.text[at++] = buf[--k];
}

// SequenceMatcher.find_longest_match(alo, ahi, blo, bhi) without junk
MIRGE_HD void longest(const char* a, const char* b, int alo, int ahi, int blo, int bhi, int& bi, int& bj, int& bk) {
    uint8_t prev[MIRGE_ISO_MAXB + 1], cur[MIRGE_ISO_MAXB + 1];
    bi = alo; bj = blo; bk = 0;
    for (int j = blo; j <= bhi; j++) prev[j] = 0;
    for (int i = alo; i < ahi; i++) {
        for (int j = blo; j < bhi; j++) {  // ascending j, as b2j lists its positions
            int k = 0;
            if (a[i] == b[j]) {
                k = (j > blo ? prev[j - 1] : 0) + 1;
                if (k > bk) { bi = i - k + 1; bj = j - k + 1; bk = k; }
            }
            cur[j] = (uint8_t)k;
        }
        for (int j = blo; j < bhi; j++) prev[j] = cur[j];
    }
}

// steps 1-3 up to the aligned lists
MIRGE_HD void aligned_lists(const char* a, int la, const char* b, int lb, Lists& L) {
    // matching blocks, explicit work list instead of difflib's queue (regions are independent of their order)
    int bi[MIRGE_ISO_MAXA + 2], bj[MIRGE_ISO_MAXA + 2], bk[MIRGE_ISO_MAXA + 2], nb = 0;
    int qa0[MIRGE_ISO_MAXA + 2], qa1[MIRGE_ISO_MAXA + 2], qb0[MIRGE_ISO_MAXA + 2], qb1[MIRGE_ISO_MAXA + 2], nq = 0;
    qa0[0] = 0; qa1[0] = la; qb0[0] = 0; qb1[0] = lb; nq = 1;
    while (nq > 0) {
        nq--;
        const int alo = qa0[nq], ahi = qa1[nq], blo = qb0[nq], bhi = qb1[nq];
        int i, j, k;
        longest(a, b, alo, ahi, blo, bhi, i, j, k);
        if (!k) continue;
        bi[nb] = i; bj[nb] = j; bk[nb] = k; nb++;
        if (alo < i && blo < j) { qa0[nq] = alo; qa1[nq] = i; qb0[nq] = blo; qb1[nq] = j; nq++; }
        if (i + k < ahi && j + k < bhi) { qa0[nq] = i + k; qa1[nq] = ahi; qb0[nq] = j + k; qb1[nq] = bhi; nq++; }
    }
    for (int x = 1; x < nb; x++) {  // matching_blocks.sort(): blocks never share a canonical position
        const int ti = bi[x], tj = bj[x], tk = bk[x];
        int y = x - 1;
        while (y >= 0 && bi[y] > ti) { bi[y + 1] = bi[y]; bj[y + 1] = bj[y]; bk[y + 1] = bk[y]; y--; }
        bi[y + 1] = ti; bj[y + 1] = tj; bk[y + 1] = tk;
    }
    bi[nb] = la; bj[nb] = lb; bk[nb] = 0; nb++;  // the closing dummy (adjacent blocks need no merging for the opcodes' text)
    int n = 0, i = 0, j = 0;
    for (int x = 0; x < nb; x++) {
        const int ai = bi[x], bjx = bj[x], size = bk[x];
        const int da = ai - i, db = bjx - j;
        // replace: the shorter side first, '-' first on a tie; delete / insert: the one side there is
        const bool plus_first = da > 0 && db > 0 && db < da;
        for (int pass = 0; pass < 2; pass++) {
            const bool plus = (pass == 0) == plus_first;
            if (plus) for (int t = j; t < bjx; t++) { L.m[n] = '-'; L.st[n] = '+'; L.sc[n] = b[t]; n++; }
            else for (int t = i; t < ai; t++) { L.m[n] = a[t]; L.st[n] = '_'; L.sc[n] = '_'; n++; }
        }
        for (int t = 0; t < size; t++) { L.m[n] = a[ai + t]; L.st[n] = '='; L.sc[n] = b[bjx + t]; n++; }
        i = ai + size; j = bjx + size;
    }
    L.n = n;
}

// `del lst[lo:hi]` / `lst.pop(k)` on the parallel arrays of one list
MIRGE_HD void del_m(Lists& L, int& nm, int lo, int hi) {
    if (lo < 0) lo = 0;
    if (hi > nm) hi = nm;
    if (hi <= lo) return;
    for (int k = hi; k < nm; k++) L.m[lo + k - hi] = L.m[k];
    nm -= hi - lo;
}
MIRGE_HD void del_s(Lists& L, int& ns, int lo, int hi) {
    if (lo < 0) lo = 0;
    if (hi > ns) hi = ns;
    if (hi <= lo) return;
    for (int k = hi; k < ns; k++) { L.st[lo + k - hi] = L.st[k]; L.sc[lo + k - hi] = L.sc[k]; }
    ns -= hi - lo;
}

// summary.py:249-289.  nm / ns: current lengths of the canonical list and of the read list (they can differ while a
// pass runs).  A Python index < 0 counts from the end; an index >= len raises IndexError, which ends that step.
MIRGE_HD void merge_replacements(Lists& L, int& nm, int& ns) {
    for (int y = 0; y < nm; y++) {  // forward: a deleted column just before an inserted one
        if (y == 0 || L.m[y] != '-') continue;
        // sub[y-1]: y-1 >= 0 here
        if (y - 1 >= ns) continue;  // IndexError
        if (L.st[y - 1] != '_') continue;
        bool two = false;
        if (y - 2 > 0) {
            if (y - 2 >= ns) continue;  // IndexError on sub[y-2]
            if (L.st[y - 2] == '_') {
                if (y + 1 >= nm) continue;  // IndexError on master_seq_bc[y+1]
                two = L.m[y + 1] == '-';
            }
        }
        if (two) { del_m(L, nm, y, y + 2); del_s(L, ns, y - 2, y); }
        else { del_m(L, nm, y, y + 1); del_s(L, ns, y - 1, y); }  // pop(y) cannot fail: y < nm, y-1 < ns
    }
    for (int y = 0; y < nm; y++) {  // reverse: a deleted column just after an inserted one
        if (y == 0 || L.m[y] != '-') continue;
        if (y + 1 >= ns) continue;  // IndexError on sub[y+1]
        if (L.st[y + 1] != '_') continue;
        bool two = false;
        if (y + 2 <= ns) {
            if (y + 2 >= ns) continue;  // sub[y+2] with y+2 == len: IndexError
            if (L.st[y + 2] == '_') {
                if (y + 1 >= nm) continue;  // IndexError on master_seq_bc[y+1]
                two = L.m[y + 1] == '-';
            }
        }
        if (two) { del_m(L, nm, y, y + 2); del_s(L, ns, y, y + 2); }
        else { del_m(L, nm, y, y + 1); del_s(L, ns, y + 1, y + 2); }
    }
}

// Python's s[lo:hi] bounds on a string of length n
MIRGE_HD void py_slice(int n, int lo, int hi, int& a, int& b) {
    if (lo < 0) { lo += n; if (lo < 0) lo = 0; }
    if (hi < 0) { hi += n; if (hi < 0) hi = 0; }
    if (lo > n) lo = n;
    if (hi > n) hi = n;
    a = lo; b = hi > lo ? hi : lo;
}

}  // namespace mirge_iso

// a / b / pre: ASCII (upper case).  start0 = precursor.find(canonical) + 1, or 1 when the precursor is "" (:181-186).
MIRGE_HD void mirge_isotype(const char* a, int la, const char* b, int lb, const char* pre, int lpre, int start0, MirgeIsoRec& o) {
    using namespace mirge_iso;
    int start = start0, end = start0 + la - 1;
    int at = 0;
    bool same = la == lb;
    for (int k = 0; same && k < la; k++) same = a[k] == b[k];
    if (same) {
        o.kind = 1; o.start = start; o.end = end;
        put(o, at, "NA"); o.vlen = (uint16_t)at;
        put_int(o, at, lb); put(o, at, "M"); o.clen = (uint16_t)(at - o.vlen);
        return;
    }
    Lists L;
    aligned_lists(a, la, b, lb, L);
    int nm = L.n, ns = L.n;
    merge_replacements(L, nm, ns);
    // :297-305 walks the canonical list and reads sub[pidx]: a shorter read list would raise IndexError out of the
    // whole per-read try block (the read is then skipped, :492); the lists stay equally long in every case seen
    if (ns < nm) { o.kind = 0; o.start = o.end = 0; o.vlen = o.clen = 0; return; }
    // ---- ends
    char s5[MIRGE_ISO_MAXN], s3r[MIRGE_ISO_MAXN];
    int n5 = 0, d5 = 0, n3 = 0, d3 = 0;
    for (int k = 0; k < nm; k++) {
        if (L.m[k] == '-') s5[n5++] = L.st[k] == '_' ? '_' : L.sc[k];
        else if (L.st[k] == '_') d5++;
        else break;
    }
    for (int k = nm; k >= 0; k--) {  // range(limit, -1, -1) looks at k-1; k = 0 looks at -1, which is in neither dict
        if (k - 1 < 0) break;
        if (L.m[k - 1] == '-') s3r[n3++] = L.st[k - 1] == '_' ? '_' : L.sc[k - 1];
        else if (L.st[k - 1] == '_') d3++;
        else break;
    }
    if (n5) {
        int lo, hi;
        py_slice(lpre, start - n5 - 1, start - 1, lo, hi);
        if (hi - lo < n5) { put(o, at, "iso_5p:-"); put_int(o, at, n5); put(o, at, ","); }
        else {
            int t = 0;
            for (int k = 0; k < n5; k++) t += s5[k] == pre[lo + k];
            if (t) { put(o, at, "iso_5p:+"); put_int(o, at, t); put(o, at, ","); }
            if (n5 - t) { put(o, at, "iso_add5p:+"); put_int(o, at, n5 - t); put(o, at, ","); }
        }
        start -= n5;
    }
    if (d5) { put(o, at, "iso_5p:+"); put_int(o, at, d5); put(o, at, ","); start += d5; }
    if (n3) {
        int lo, hi;
        py_slice(lpre, end, end + n3, lo, hi);
        if (hi - lo < n3) { put(o, at, "iso_3p:+"); put_int(o, at, n3); put(o, at, ","); }
        else {
            int t = 0;
            for (int k = 0; k < n3; k++) t += s3r[n3 - 1 - k] == pre[lo + k];
            if (t) { put(o, at, "iso_3p:+"); put_int(o, at, t); put(o, at, ","); }
            if (n3 - t) { put(o, at, "iso_add3p:+"); put_int(o, at, n3 - t); put(o, at, ","); }
        }
        end += n3;
    }
    if (d3) { put(o, at, "iso_3p:-"); put_int(o, at, d3); put(o, at, ","); end -= d3; }
    // ---- substitutions by column index, classes in order of first appearance (:404-418)
    bool seen[5] = {false, false, false, false, false};
    for (int k = 0; k < nm; k++) {
        if (L.m[k] == '-' || L.st[k] == '_') continue;
        if (L.st[k] == '=' && L.sc[k] == L.m[k]) continue;
        const int c = k == 7 ? 0 : (k >= 1 && k <= 6) ? 1 : (k >= 8 && k <= 12) ? 2 : (k >= 13 && k <= 17) ? 3 : 4;
        if (seen[c]) continue;
        seen[c] = true;
        put(o, at, c == 0 ? "iso_snv_central_offset," : c == 1 ? "iso_snv_seed," : c == 2 ? "iso_snv_central," : c == 3 ? "iso_snv_central_supp," : "iso_snv,");
    }
    if (at > 0 && o.text[at - 1] == ',') at--;
    if (at == 0) put(o, at, "iso_snv");
    o.vlen = (uint16_t)at;
    // ---- CIGAR (:427-463)
    char cs[MIRGE_ISO_MAXN + 2];
    bool any_base = false;
    for (int k = 0; k < nm; k++) {
        const bool M = (L.st[k] == '=' && L.sc[k] == L.m[k]) || L.m[k] == '-' || L.st[k] == '_';
        cs[k] = M ? 'M' : L.m[k];
        any_base |= cs[k] == 'A' || cs[k] == 'T' || cs[k] == 'G' || cs[k] == 'C';
    }
    if (!any_base) { put_int(o, at, lb); put(o, at, "M"); }
    else {
        int run = 0;
        for (int k = 0; k < nm; k++) {
            if (k != 0 && cs[k] != cs[k - 1]) {
                if (run != 1) put_int(o, at, run);
                if (at < MIRGE_ISO_TEXT - 1) o.text[at++] = cs[k - 1];
                run = 1;
            } else run++;
        }
        if (run != 1) put_int(o, at, run);
        if (nm > 0 && at < MIRGE_ISO_TEXT - 1) o.text[at++] = cs[nm - 1];
    }
    o.clen = (uint16_t)(at - o.vlen);
    o.kind = 2; o.start = start; o.end = end;
}

// ---------------------------------------------------------------------------------------------------------------------
// The same record WITHOUT per-thread arrays (round 5).  mirge_isotype above keeps its DP rows, block lists, aligned lists and
// text in ~2 KB of per-thread scratch: on the GPU every element is a memory access of its own and a wave spends its time
// waiting for them (11.8 ms per 0.72 M reads).  Here everything lives in registers:
//   * a sequence is three bit planes of its letters' codes (A C G T N = 0..4), bit p = position p;
//   * find_longest_match walks the DIAGONALS of the comparison: a[i] == b[i + d] for all i at once is five AND / shift pairs,
//     the longest run of ones and its first start two instructions per base of the run;
//   * the two aligned lists are bit planes over their columns (gap / canonical code / deleted / inserted / read code); Python's
//     `del lst[lo:hi]` is a shift of the planes' upper part;
//   * the text goes straight to where the record lives.
// Only the matching blocks and the work list of the recursion are indexed at run time: two small arrays behind `WS` (LDS on
// the device, the stack on the host).  Same difflib order, same tie-breaks, same list-iterator quirks; the function above
// stays as the general path (letters beyond ACGTN, more than 64 columns, more than MIRGE_ISO_FAST_BLOCKS blocks) and as the
// reference the tests fuzz this one against (tests/hostsim, millions of pairs).
#define MIRGE_ISO_FAST_BLOCKS 16

namespace mirge_iso {

MIRGE_HD uint64_t lowmask(int n) { return n >= 64 ? ~0ull : n <= 0 ? 0ull : ((1ull << n) - 1ull); }
MIRGE_HD int popc64(uint64_t x) { return __builtin_popcountll(x); }
MIRGE_HD int ctz64(uint64_t x) { return __builtin_ctzll(x); }
MIRGE_HD int clz64(uint64_t x) { return __builtin_clzll(x); }

struct Seq {           // letters as code bit planes
    uint64_t c0, c1, c2;
    int n;
    MIRGE_HD int code(int p) const { return (int)((c0 >> p) & 1ull) | (int)(((c1 >> p) & 1ull) << 1) | (int)(((c2 >> p) & 1ull) << 2); }
    MIRGE_HD uint64_t letter(int c) const {  // positions holding letter c
        const uint64_t v = lowmask(n);
        return v & ((c & 1) ? c0 : ~c0) & ((c & 2) ? c1 : ~c1) & ((c & 4) ? c2 : ~c2);
    }
};
MIRGE_HD int iso_code(char ch) { return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : ch == 'N' ? 4 : -1; }
MIRGE_HD char iso_char(int code) { return code == 0 ? 'A' : code == 1 ? 'C' : code == 2 ? 'G' : code == 3 ? 'T' : 'N'; }
MIRGE_HD bool seq_of_ascii(const char* s, int n, Seq& q) {
    q.c0 = q.c1 = q.c2 = 0; q.n = n;
    if (n > 64) return false;
    bool ok = true;
    for (int p = 0; p < n; p++) {
        const int c = iso_code(s[p]);
        ok &= c >= 0;
        q.c0 |= (uint64_t)(c & 1) << p; q.c1 |= (uint64_t)((c >> 1) & 1) << p; q.c2 |= (uint64_t)((c >> 2) & 1) << p;
    }
    return ok;
}

struct Letters { uint64_t m[5]; };  // (indexed with compile-time constants only)
MIRGE_HD void letters_of(const Seq& q, Letters& L) {
    L.m[0] = q.letter(0); L.m[1] = q.letter(1); L.m[2] = q.letter(2); L.m[3] = q.letter(3); L.m[4] = q.letter(4);
}

// SequenceMatcher.find_longest_match(alo, ahi, blo, bhi): the longest block, then the earliest in a, then the earliest in b
MIRGE_HD void longest_fast(const Letters& A, const Letters& B, bool any_n, int alo, int ahi, int blo, int bhi, int& bi, int& bj, int& bk) {
    bi = alo; bj = blo; bk = 0;
    if (alo >= ahi || blo >= bhi) return;
    for (int d = blo - (ahi - 1); d <= (bhi - 1) - alo; d++) {  // diagonal j = i + d
        const int lo = alo > blo - d ? alo : blo - d, hi = ahi < bhi - d ? ahi : bhi - d;
        uint64_t M;
        if (d >= 0) {
            M = (A.m[0] & (B.m[0] >> d)) | (A.m[1] & (B.m[1] >> d)) | (A.m[2] & (B.m[2] >> d)) | (A.m[3] & (B.m[3] >> d));
            if (any_n) M |= A.m[4] & (B.m[4] >> d);
        } else {
            M = (A.m[0] & (B.m[0] << -d)) | (A.m[1] & (B.m[1] << -d)) | (A.m[2] & (B.m[2] << -d)) | (A.m[3] & (B.m[3] << -d));
            if (any_n) M |= A.m[4] & (B.m[4] << -d);
        }
        M &= lowmask(hi) & ~lowmask(lo);
        if (!M) continue;
        uint64_t r = M, last;
        int k = 0;
        do { last = r; r &= r >> 1; k++; } while (r);  // `last`: starts of the runs of the maximal length k
        if (k < bk) continue;
        const int i = ctz64(last), j = i + d;
        if (k > bk || i < bi || (i == bi && j < bj)) { bi = i; bj = j; bk = k; }
    }
}

struct Cols {  // the two aligned lists, one bit per column
    uint64_t gap, m0, m1, m2;       // canonical list: '-' or the canonical's letter
    uint64_t del, ins, s0, s1, s2;  // read list: '_' deleted, '+' inserted (else equal), the read's letter
    MIRGE_HD int mcode(int k) const { return (int)((m0 >> k) & 1ull) | (int)(((m1 >> k) & 1ull) << 1) | (int)(((m2 >> k) & 1ull) << 2); }
    MIRGE_HD int scode(int k) const { return (int)((s0 >> k) & 1ull) | (int)(((s1 >> k) & 1ull) << 1) | (int)(((s2 >> k) & 1ull) << 2); }
};
MIRGE_HD uint64_t cut(uint64_t x, int lo, int w) { return (x & lowmask(lo)) | ((x >> w) & ~lowmask(lo)); }  // del x[lo:lo+w]
MIRGE_HD void del_m_fast(Cols& L, int& nm, int lo, int hi) {
    if (lo < 0) lo = 0;
    if (hi > nm) hi = nm;
    if (hi <= lo) return;
    const int w = hi - lo;
    L.gap = cut(L.gap, lo, w); L.m0 = cut(L.m0, lo, w); L.m1 = cut(L.m1, lo, w); L.m2 = cut(L.m2, lo, w);
    nm -= w;
}
MIRGE_HD void del_s_fast(Cols& L, int& ns, int lo, int hi) {
    if (lo < 0) lo = 0;
    if (hi > ns) hi = ns;
    if (hi <= lo) return;
    const int w = hi - lo;
    L.del = cut(L.del, lo, w); L.ins = cut(L.ins, lo, w); L.s0 = cut(L.s0, lo, w); L.s1 = cut(L.s1, lo, w); L.s2 = cut(L.s2, lo, w);
    ns -= w;
}
MIRGE_HD uint64_t field(uint64_t x, int from, int w, int to) { return ((x >> from) & lowmask(w)) << to; }

// merge_replacements on the planes: only '-' columns of the canonical list do anything, so the walk jumps from one to the next
MIRGE_HD void merge_replacements_fast(Cols& L, int& nm, int& ns) {
    for (int pass = 0; pass < 2; pass++) {
        int y = 0;
        for (;;) {
            const uint64_t g = L.gap & lowmask(nm) & ~lowmask(y);
            if (!g) break;
            y = ctz64(g);
            const int here = y;
            y++;  // (the iterator moves on whatever happens to the lists)
            if (here == 0) continue;
            if (pass == 0) {
                if (here - 1 >= ns) continue;
                if (!((L.del >> (here - 1)) & 1ull)) continue;
                bool two = false;
                if (here - 2 > 0) {
                    if (here - 2 >= ns) continue;
                    if ((L.del >> (here - 2)) & 1ull) {
                        if (here + 1 >= nm) continue;
                        two = (L.gap >> (here + 1)) & 1ull;
                    }
                }
                if (two) { del_m_fast(L, nm, here, here + 2); del_s_fast(L, ns, here - 2, here); }
                else { del_m_fast(L, nm, here, here + 1); del_s_fast(L, ns, here - 1, here); }
            } else {
                if (here + 1 >= ns) continue;
                if (!((L.del >> (here + 1)) & 1ull)) continue;
                bool two = false;
                if (here + 2 <= ns) {
                    if (here + 2 >= ns) continue;
                    if ((L.del >> (here + 2)) & 1ull) {
                        if (here + 1 >= nm) continue;
                        two = (L.gap >> (here + 1)) & 1ull;
                    }
                }
                if (two) { del_m_fast(L, nm, here, here + 2); del_s_fast(L, ns, here, here + 2); }
                else { del_m_fast(L, nm, here, here + 1); del_s_fast(L, ns, here + 1, here + 2); }
            }
        }
    }
}

struct TextOut {  // the record's text, written where it lives
    char* t;
    int at;
    char last;
    MIRGE_HD void ch(char c) { if (at < MIRGE_ISO_TEXT - 1) { t[at++] = c; last = c; } }
    MIRGE_HD void str(const char* s) { for (int k = 0; s[k]; k++) ch(s[k]); }
    MIRGE_HD void num(int v) {
        if (v < 0) { ch('-'); v = -v; }
        int div = 1;
        while (v / div >= 10) div *= 10;
        for (; div > 0; div /= 10) ch((char)('0' + (v / div) % 10));
    }
};

}  // namespace mirge_iso

// WS: `uint32_t& blk(int k)`, `uint32_t& que(int k)`, k < MIRGE_ISO_FAST_BLOCKS each.  Returns false -- nothing written -- when the
// pair is not one for this path; *kind / *start / *end / *vlen / *clen and text[] are the record's fields otherwise.
template <class WS>
MIRGE_HD bool mirge_isotype_fast(const mirge_iso::Seq& a, const mirge_iso::Seq& b, const char* pre, int lpre, int start0, WS& ws,
                                 int32_t* o_start, int32_t* o_end, uint8_t* o_kind, uint16_t* o_vlen, uint16_t* o_clen, char* text) {
    using namespace mirge_iso;
    const int la = a.n, lb = b.n;
    if (la > MIRGE_ISO_MAXA || lb > MIRGE_ISO_MAXB || la + lb > 64 || la < 1 || lb < 1) return false;
    int start = start0, end = start0 + la - 1;
    TextOut o;
    o.t = text; o.at = 0; o.last = 0;
    if (la == lb && a.c0 == b.c0 && a.c1 == b.c1 && a.c2 == b.c2) {
        o.str("NA");
        const int v = o.at;
        o.num(lb); o.ch('M');
        *o_kind = 1; *o_start = start; *o_end = end; *o_vlen = (uint16_t)v; *o_clen = (uint16_t)(o.at - v);
        return true;
    }
    // ---- matching blocks (get_matching_blocks), the recursion as a work list
    Letters A, B;
    letters_of(a, A); letters_of(b, B);
    const bool any_n = (A.m[4] | B.m[4]) != 0;
    int nb = 0, nq = 1;
    ws.que(0) = 0u | ((uint32_t)la << 8) | (0u << 16) | ((uint32_t)lb << 24);
    while (nq > 0) {
        nq--;
        const uint32_t q = ws.que(nq);
        const int alo = q & 255, ahi = (q >> 8) & 255, blo = (q >> 16) & 255, bhi = q >> 24;
        int i, j, k;
        longest_fast(A, B, any_n, alo, ahi, blo, bhi, i, j, k);
        if (!k) continue;
        if (nb >= MIRGE_ISO_FAST_BLOCKS || nq + 2 > MIRGE_ISO_FAST_BLOCKS) return false;
        ws.blk(nb++) = (uint32_t)i | ((uint32_t)j << 8) | ((uint32_t)k << 16);
        if (alo < i && blo < j) ws.que(nq++) = (uint32_t)alo | ((uint32_t)i << 8) | ((uint32_t)blo << 16) | ((uint32_t)j << 24);
        if (i + k < ahi && j + k < bhi) ws.que(nq++) = (uint32_t)(i + k) | ((uint32_t)ahi << 8) | ((uint32_t)(j + k) << 16) | ((uint32_t)bhi << 24);
    }
    for (int x = 1; x < nb; x++) {  // sort by position in the canonical (blocks never share one)
        const uint32_t t = ws.blk(x);
        int y = x - 1;
        while (y >= 0 && (ws.blk(y) & 255u) > (t & 255u)) { ws.blk(y + 1) = ws.blk(y); y--; }
        ws.blk(y + 1) = t;
    }
    // ---- the aligned lists, a run of columns at a time
    Cols L;
    L.gap = L.m0 = L.m1 = L.m2 = L.del = L.ins = L.s0 = L.s1 = L.s2 = 0;
    int n = 0, i = 0, j = 0;
    for (int x = 0; x <= nb; x++) {
        const uint32_t blkv = x < nb ? ws.blk(x) : ((uint32_t)la | ((uint32_t)lb << 8));
        const int ai = blkv & 255, bjx = (blkv >> 8) & 255, size = (blkv >> 16) & 255;
        const int da = ai - i, db = bjx - j;
        const bool plus_first = da > 0 && db > 0 && db < da;
        for (int pass = 0; pass < 2; pass++) {
            const bool plus = (pass == 0) == plus_first;
            if (plus) {
                if (db > 0) {
                    L.gap |= lowmask(db) << n; L.ins |= lowmask(db) << n;
                    L.s0 |= field(b.c0, j, db, n); L.s1 |= field(b.c1, j, db, n); L.s2 |= field(b.c2, j, db, n);
                    n += db;
                }
            } else if (da > 0) {
                L.del |= lowmask(da) << n;
                L.m0 |= field(a.c0, i, da, n); L.m1 |= field(a.c1, i, da, n); L.m2 |= field(a.c2, i, da, n);
                n += da;
            }
        }
        if (size > 0) {
            L.m0 |= field(a.c0, ai, size, n); L.m1 |= field(a.c1, ai, size, n); L.m2 |= field(a.c2, ai, size, n);
            L.s0 |= field(b.c0, bjx, size, n); L.s1 |= field(b.c1, bjx, size, n); L.s2 |= field(b.c2, bjx, size, n);
            n += size;
        }
        i = ai + size; j = bjx + size;
    }
    int nm = n, ns = n;
    merge_replacements_fast(L, nm, ns);
    if (ns < nm) { *o_kind = 0; *o_start = 0; *o_end = 0; *o_vlen = 0; *o_clen = 0; return true; }
    const uint64_t valid = lowmask(nm);
    const uint64_t gap = L.gap & valid, del = L.del & valid;
    // ---- ends: the leading / trailing columns that are '-' in the canonical list or deleted in the read list
    const uint64_t body = ~(gap | del) & valid;
    const int lead = body ? ctz64(body) : nm;
    const int tail0 = body ? 64 - clz64(body) : 0;  // first column of the trailing run
    const uint64_t lead_m = lowmask(lead), tail_m = valid & ~lowmask(tail0);
    const int n5 = popc64(gap & lead_m), d5 = popc64(del & ~gap & lead_m);
    const int n3 = popc64(gap & tail_m), d3 = popc64(del & ~gap & tail_m);
    if (n5) {
        int lo, hi;
        py_slice(lpre, start - n5 - 1, start - 1, lo, hi);
        if (hi - lo < n5) { o.str("iso_5p:-"); o.num(n5); o.ch(','); }
        else {
            int t = 0, idx = 0;
            for (uint64_t g = gap & lead_m; g; g &= g - 1, idx++) {
                const int k = ctz64(g);
                if (!((del >> k) & 1ull)) t += iso_char(L.scode(k)) == pre[lo + idx];
            }
            if (t) { o.str("iso_5p:+"); o.num(t); o.ch(','); }
            if (n5 - t) { o.str("iso_add5p:+"); o.num(n5 - t); o.ch(','); }
        }
        start -= n5;
    }
    if (d5) { o.str("iso_5p:+"); o.num(d5); o.ch(','); start += d5; }
    if (n3) {
        int lo, hi;
        py_slice(lpre, end, end + n3, lo, hi);
        if (hi - lo < n3) { o.str("iso_3p:+"); o.num(n3); o.ch(','); }
        else {
            int t = 0, idx = 0;
            for (uint64_t g = gap & tail_m; g; g &= g - 1, idx++) {
                const int k = ctz64(g);
                if (!((del >> k) & 1ull)) t += iso_char(L.scode(k)) == pre[lo + idx];
            }
            if (t) { o.str("iso_3p:+"); o.num(t); o.ch(','); }
            if (n3 - t) { o.str("iso_add3p:+"); o.num(n3 - t); o.ch(','); }
        }
        end += n3;
    }
    if (d3) { o.str("iso_3p:-"); o.num(d3); o.ch(','); end -= d3; }
    // ---- substitutions: a column with a letter on both sides that is not an equal pair
    const uint64_t differ = (L.m0 ^ L.s0) | (L.m1 ^ L.s1) | (L.m2 ^ L.s2);
    const uint64_t sub = ~gap & ~del & valid & (L.ins | differ);
    bool seen[5] = {false, false, false, false, false};
    for (uint64_t g = sub; g; g &= g - 1) {
        const int k = ctz64(g);
        const int c = k == 7 ? 0 : (k >= 1 && k <= 6) ? 1 : (k >= 8 && k <= 12) ? 2 : (k >= 13 && k <= 17) ? 3 : 4;
        const bool was = c == 0 ? seen[0] : c == 1 ? seen[1] : c == 2 ? seen[2] : c == 3 ? seen[3] : seen[4];
        if (was) continue;
        if (c == 0) seen[0] = true; else if (c == 1) seen[1] = true; else if (c == 2) seen[2] = true; else if (c == 3) seen[3] = true; else seen[4] = true;
        o.str(c == 0 ? "iso_snv_central_offset," : c == 1 ? "iso_snv_seed," : c == 2 ? "iso_snv_central," : c == 3 ? "iso_snv_central_supp," : "iso_snv,");
    }
    if (o.at > 0 && o.last == ',') o.at--;
    if (o.at == 0) o.str("iso_snv");
    const int vlen = o.at;
    // ---- CIGAR: 'M' for every column that is no substitution, the canonical's letter for one
    const uint64_t any_base = sub & ~L.m2;  // (a substituted 'N' of the canonical prints as N and does not count, as in :427-463)
    if (!any_base) { o.num(lb); o.ch('M'); }
    else {
        int run = 0;
        char prev = 0;
        for (int k = 0; k < nm; k++) {
            const char cs = ((sub >> k) & 1ull) ? iso_char(L.mcode(k)) : 'M';
            if (k != 0 && cs != prev) {
                if (run != 1) o.num(run);
                o.ch(prev);
                run = 1;
            } else run++;
            prev = cs;
        }
        if (run != 1) o.num(run);
        if (nm > 0) o.ch(prev);
    }
    *o_kind = 2; *o_start = start; *o_end = end; *o_vlen = (uint16_t)vlen; *o_clen = (uint16_t)(o.at - vlen);
    return true;
}
