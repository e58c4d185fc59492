"""Minimal writer of bowtie-1 index files -- TEST INFRASTRUCTURE for mirge3_amd.ebwt.

No bowtie-build exists in this image, so the reader is exercised against files laid out as the bowtie 1.x sources
lay them out (``Ebwt::writeFromMemory``, ``EbwtParams``, ``BitPairReference`` / ``RefRecord``): ``.1.ebwt`` = header,
reference lengths, fragment table, the BWT block (here: zero bytes of the right size -- nothing on this path reads
it), zOff, fchr, ftab, eftab, then the names; ``.3.ebwt`` = (off, len, first) records; ``.4.ebwt`` = 2-bit bases;
``.2.ebwt`` = the sampled suffix array (zeros).  Only what ``bowtie-inspect`` needs to print names and sequences is real.
"""
import struct

import numpy as np


def write_ebwt(base, headers, seqs, large=False, line_rate=6, off_rate=5, ftab_chars=10, big_endian=False):
    en = ">" if big_endian else "<"
    osz, o = (8, "Q") if large else (4, "I")
    ext = ".ebwtl" if large else ".ebwt"
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    recs, bases, plen = [], [], []
    for s in seqs:
        s = s.upper().replace("U", "T")
        i, first, unamb = 0, True, 0
        if not s:
            recs.append((0, 0, True))
        while i < len(s):
            j = i
            while j < len(s) and s[j] not in code:
                j += 1
            k = j
            while k < len(s) and s[k] in code:
                k += 1
            recs.append((j - i, k - j, first))
            first = False
            bases.extend(code[c] for c in s[j:k])
            unamb += k - j
            i = k
        plen.append(unamb)
    total = len(bases)
    with open(base + ".3" + ext, "wb") as fh:
        fh.write(struct.pack(en + "i", 1) + struct.pack(en + o, len(recs)))
        for off, ln, first in recs:
            fh.write(struct.pack(en + o + o, off, ln) + (b"\x01" if first else b"\x00"))
    b = np.zeros((total + 3) // 4 * 4, dtype=np.uint8)
    b[:total] = bases
    packed = (b[0::4] | (b[1::4] << 2) | (b[2::4] << 4) | (b[3::4] << 6)).astype(np.uint8)
    packed.tofile(base + ".4" + ext)
    # .1.ebwt
    n_frag = sum(1 for _, ln, _ in recs if ln > 0)
    bwt_sz = total // 4 + 1
    side_sz = 1 << line_rate
    side_bwt_sz = side_sz - 2 * osz
    n_side_pairs = (bwt_sz + 2 * side_bwt_sz - 1) // (2 * side_bwt_sz)
    ebwt_tot = n_side_pairs * 2 * side_sz
    ftab_len, eftab_len = (1 << (2 * ftab_chars)) + 1, 2 * ftab_chars
    with open(base + ".1" + ext, "wb") as fh:
        fh.write(struct.pack(en + "i", 1) + struct.pack(en + o, total))
        fh.write(struct.pack(en + "5i", line_rate, 1, off_rate, ftab_chars, -(1 | 4)))  # flags: new style, entireReverse
        fh.write(struct.pack(en + o, len(seqs)))
        fh.write(b"".join(struct.pack(en + o, x) for x in plen))
        fh.write(struct.pack(en + o, n_frag))
        joined = 0
        for sid, s in enumerate(seqs):
            pass
        # fragment table: (offset in the joined text, reference id, offset in the reference) per unambiguous stretch
        sid, in_ref = -1, 0
        for off, ln, first in recs:
            if first:
                sid += 1
                in_ref = 0
            in_ref += off
            if ln > 0:
                fh.write(struct.pack(en + o + o + o, joined, sid, in_ref))
            joined += ln
            in_ref += ln
        fh.write(bytes(ebwt_tot))
        fh.write(struct.pack(en + o, 0))                      # zOff
        fh.write(bytes(osz * (5 + ftab_len + eftab_len)))     # fchr, ftab, eftab
        names_at = fh.tell()
        for h in headers:
            fh.write(h.encode("ascii") + b"\n")
        fh.write(b"\0")
    with open(base + ".2" + ext, "wb") as fh:
        fh.write(struct.pack(en + "i", 1))
        fh.write(bytes(osz * ((total + 1 + (1 << off_rate) - 1) >> off_rate)))
    return dict(names_offset=names_at, n_frag=n_frag, n_records=len(recs))


def fasta_dir_to_ebwt(index_dir, remove_fasta=True, **kw):
    """turn every <name>.fa of a library directory into <name>.{1,2,3,4}.ebwt (what a miRge3.0 library ships)"""
    import os
    import mirge3_amd  # noqa: F401
    from mirge3_amd.seqio import read_fasta
    for f in sorted(os.listdir(index_dir)):
        if not f.endswith(".fa"):
            continue
        base = os.path.join(index_dir, f[:-3])
        lib = read_fasta(base + ".fa")
        write_ebwt(base, lib.headers, lib.seqs.to_list(), **kw)
        if remove_fasta:
            os.remove(base + ".fa")
