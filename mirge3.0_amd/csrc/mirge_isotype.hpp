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
