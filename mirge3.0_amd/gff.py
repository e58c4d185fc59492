"""miRTop GFF3 (``sample_miRge3.gff``; ``-gff``) -- host mirror of the GFF block of ``summarize`` and of ``create_gff``
(``mirge/libs/summary.py:48-606,800-837``).

The reference types every miRNA read on the host: ``difflib.Differ`` between the miRNA's canonical sequence and the
read, two passes of in-place list rewriting, then string assembly -- its largest host cost (71 s against 2 s for the rest
in ``docs/source/quick_start.md:117-158``).  Here the typing is one kernel (``k_isotype`` / ``mirge_isomir_type``: the same
diff and the same rewrites per read, ``csrc/mirge_isotype.hpp``) and the file is assembled by ``mirge_gff_write`` on the
host's cores.  What stays in Python is per miRNA NAME, a few thousand entries: which canonical sequence, which
precursor, where the canonical sits in it (``resolve_names``) -- read exactly the way the reference reads them,
including its quirk that the LAST precursor of the hairpin index has an empty sequence (``bowtie-inspect``'s output
ends with a newline and the parser assigns the trailing '' to it, summary.py:819-826).

The interactive report's JavaScript series (``html_data.*``, summary.py:505-606) are outside the hot path.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np

from . import _ffi
from .cascade import Cascade, EXACT_PASS, ISO_PASS
from .seqio import FlatSeqs

RECORD = np.dtype([("start", "<i4"), ("end", "<i4"), ("kind", "u1"), ("pad", "u1"), ("vlen", "<u2"), ("clen", "<u2"),
                   ("text", "S320")], align=True)
assert RECORD.itemsize == 336


def read_mature_fasta(path) -> Dict[str, str]:
    """summary.py:827-836: a line with '>' names the entry, any other line is its sequence (the last one wins)"""
    out, name = {}, None
    with open(path) as fh:
        for ln in fh:
            ln = ln.strip()
            if '>' in ln:
                name = ln.replace(">", "")
            else:
                out[name] = ln
    return out


def precursor_dict(hairpin) -> Dict[str, str]:
    """summary.py:813-826 on what ``bowtie-inspect -a 20000 -e <hairpin index>`` prints: '>header' then the sequence
    on one line, and a final newline -- whose empty remainder overwrites the last precursor's sequence"""
    out, name = {}, None
    for h, s in zip(hairpin.headers, hairpin.seqs.to_list()):
        name = h.split(" ")[0]
        for k in range(0, max(len(s), 1), 20000):
            out[name] = s[k:k + 20000]
    if name is not None:
        out[name] = ""
    return out


def read_annotation(path, ref_db: str) -> Dict[str, str]:
    """mature name -> precursor name from ``<org>_<db>.gff3`` (summary.py:96-128; first mention of a mature wins)"""
    pre_of: Dict[str, str] = {}
    pre_name = None
    with open(path) as fh:
        for ln in fh:
            f = ln.strip().split("\t")
            try:
                if ref_db == "MirGeneDB":
                    if f[2] == "pre_miRNA":
                        pre_name = f[8].split(";")[0].replace("ID=", "")
                    else:
                        pre_of.setdefault(f[8].split(";")[0].replace("ID=", ""), pre_name)
                else:
                    if f[2] == "miRNA_primary_transcript":
                        pre_name = f[8].split(";")[-1].replace("Name=", "")
                    else:
                        pre_of.setdefault(f[8].split(";")[2].replace("Name=", ""), pre_name)
            except IndexError:
                pass
    return pre_of


def resolve_names(mirna_names: List[str], mirDict: Dict[str, str], pre_of: Dict[str, str], pre_mirDict: Dict[str, str]):
    """Per reference of the miRNA library, what create_gff (:136-186) looks up from its NAME: the name it prints
    (``.SNP`` suffix cut; ``-3p`` / ``-5p`` cut when the full name has no annotation), the canonical sequence, the
    precursor and the canonical's 1-based position in it.  A name the reference would drop (KeyError -> pass, :492)
    gets master -1."""
    printed: List[str] = []
    pidx: Dict[str, int] = {}
    parents: List[str] = []
    paridx: Dict[str, int] = {}
    masters: List[str] = []
    midx: Dict[tuple, int] = {}
    pre_seqs: List[str] = []
    pre_of_master: List[int] = []
    start0: List[int] = []
    master_of_ref = np.full(len(mirna_names), -1, dtype=np.int32)
    name_of_ref = np.full(len(mirna_names), -1, dtype=np.int32)
    parent_of_ref = np.full(len(mirna_names), -1, dtype=np.int32)
    for r, nm in enumerate(mirna_names):
        sm = nm.split(".")[0] if "." in nm else nm
        if sm not in pre_of:
            sm = sm.replace("-3p", "").replace("-5p", "").replace("-3p*", "").replace("-5p*", "")
        if sm not in mirDict or sm not in pre_of or pre_of[sm] not in pre_mirDict:
            continue
        master, parent = mirDict[sm], pre_of[sm]
        pseq = pre_mirDict[parent]
        key = (master, parent)
        if key not in midx:
            if parent not in paridx:
                paridx[parent] = len(parents)
                parents.append(parent)
                pre_seqs.append(pseq)
            midx[key] = len(masters)
            masters.append(master)
            pre_of_master.append(paridx[parent])
            start0.append((pseq.find(master) + 1) if pseq != "" else 1)
        master_of_ref[r] = midx[key]
        name_of_ref[r] = pidx.setdefault(sm, len(pidx))
        if len(printed) < len(pidx):
            printed.append(sm)
        parent_of_ref[r] = paridx[parent]
    return dict(master_of_ref=master_of_ref, name_of_ref=name_of_ref, parent_of_ref=parent_of_ref, printed=printed,
                parents=parents, masters=masters, pre_seqs=pre_seqs, pre_of_master=np.asarray(pre_of_master, dtype=np.int32),
                start0=np.asarray(start0, dtype=np.int32))


def isomir_records(casc: Cascade, uniq, res, tables: dict, rows: np.ndarray) -> np.ndarray:
    """``mirge_isomir_type`` for the reads ``rows`` (handle indices, in print order) -> structured array of records"""
    lib = _ffi.load()
    slot = np.full(len(uniq), -1, dtype=np.int32)
    slot[rows] = np.arange(rows.shape[0], dtype=np.int32)
    m = FlatSeqs.from_list(tables["masters"])
    p = FlatSeqs.from_list(tables["pre_seqs"])
    moff, poff = m.offsets.astype(np.int32), p.offsets.astype(np.int32)
    mdata = np.ascontiguousarray(m.data) if m.data.size else np.zeros(1, np.uint8)
    pdata = np.ascontiguousarray(p.data) if p.data.size else np.zeros(1, np.uint8)
    out = np.zeros(max(rows.shape[0], 1), dtype=RECORD)
    mof = np.ascontiguousarray(tables["master_of_ref"], dtype=np.int32)
    pom = np.ascontiguousarray(tables["pre_of_master"], dtype=np.int32) if len(m) else np.zeros(1, np.int32)
    s0 = np.ascontiguousarray(tables["start0"], dtype=np.int32) if len(m) else np.zeros(1, np.int32)
    _ffi._check(lib.mirge_isomir_type(casc.ctx._h, uniq._h, res._h, C.c_int32(EXACT_PASS), C.c_int32(ISO_PASS), _ffi._p(mof),
                                      C.c_int64(mof.shape[0]), _ffi._p(mdata), _ffi._p(moff), _ffi._p(pom), _ffi._p(s0),
                                      C.c_int64(len(m)), _ffi._p(pdata), _ffi._p(poff), C.c_int64(len(p)), _ffi._p(slot),
                                      C.c_int64(rows.shape[0]), _ffi._p(out)), "mirge_isomir_type")
    return out[:rows.shape[0]]


def device_route() -> bool:
    """the GFF3 formatted on the device (``mirge_gff_write_device``, round 6); MIRGE_GFF_DEVICE=0: records to the host and
    ``mirge_gff_write`` on its cores (round 5's route -- the tests' second implementation)"""
    import os
    return os.environ.get("MIRGE_GFF_DEVICE", "1") != "0"


def name_tables(args, ref_db, casc: Cascade) -> dict:
    """``resolve_names`` of the run's libraries, kept with the cascade: a few thousand names, read once per process"""
    key = (str(args.libraries_path), args.organism_name, ref_db)
    cached = getattr(casc, "_gff_tables", None)
    if cached is None or cached[0] != key:
        lp, org = Path(args.libraries_path), args.organism_name
        mirDict = read_mature_fasta(lp / org / "fasta.Libs" / (org + "_mature_" + ref_db + ".fa"))
        pre_of = read_annotation(lp / org / "annotation.Libs" / (org + "_" + ref_db + ".gff3"), ref_db)
        cached = (key, resolve_names(casc.libs["mirna"].names, mirDict, pre_of, precursor_dict(casc.libs["hairpin"])))
        casc._gff_tables = cached
    return cached[1]


def write_gff_device(args, workDir, ref_db, base_names, casc: Cascade, uniq, res, order):
    """``-gff`` from the device-resident run: rows chosen, typed and formatted on the GPU (``mirge_gff_write_device``); nothing per
    read is fetched.  ``order`` = the frame's row order (handle indices)."""
    import time
    t0 = time.perf_counter()
    tables = name_tables(args, ref_db, casc)
    t_names = time.perf_counter() - t0
    out = write_gff_device_with(tables, Path(workDir) / "sample_miRge3.gff", ref_db, base_names, casc, uniq, res, order)
    out["timing"]["name_tables_s"] = round(t_names, 4)
    return out


def write_gff_device_with(tables: dict, path, ref_db, base_names, casc: Cascade, uniq, res, order):
    """``mirge_gff_write_device`` with the name tables given (``resolve_names``)"""
    import time
    tm = {}
    t0 = time.perf_counter()
    version_db = "miRBase22" if ref_db == "miRBase" else "MirGeneDB2.0"
    head = ("# GFF3 adapted for miRNA sequencing data\n## VERSION 0.0.1\n## source-ontology: " + version_db + "\n## COLDATA: " +
            ",".join(str(nm) for nm in base_names) + "\n")
    m = FlatSeqs.from_list(tables["masters"])
    p = FlatSeqs.from_list(tables["pre_seqs"])
    names, parents = FlatSeqs.from_list(tables["printed"]), FlatSeqs.from_list(tables["parents"])
    z8, z32 = np.zeros(1, np.uint8), np.zeros(1, np.int32)
    moff, poff = m.offsets.astype(np.int32), p.offsets.astype(np.int32)
    mdata = np.ascontiguousarray(m.data) if m.data.size else z8
    pdata = np.ascontiguousarray(p.data) if p.data.size else z8
    mof = np.ascontiguousarray(tables["master_of_ref"], dtype=np.int32)
    pom = np.ascontiguousarray(tables["pre_of_master"], dtype=np.int32) if len(m) else z32
    s0 = np.ascontiguousarray(tables["start0"], dtype=np.int32) if len(m) else z32
    nof = np.ascontiguousarray(tables["name_of_ref"], dtype=np.int32)
    pof = np.ascontiguousarray(tables["parent_of_ref"], dtype=np.int32)
    nd = np.ascontiguousarray(names.data) if names.data.size else z8
    pd_ = np.ascontiguousarray(parents.data) if parents.data.size else z8
    noff = np.ascontiguousarray(names.offsets, dtype=np.int64)
    paroff = np.ascontiguousarray(parents.offsets, dtype=np.int64)
    order = np.ascontiguousarray(order, dtype=np.int64)
    n_lines = C.c_int64(0)
    _ffi._check(_ffi.load().mirge_gff_write_device(
        casc.ctx._h, uniq._h, res._h, C.c_int32(EXACT_PASS), C.c_int32(ISO_PASS), _ffi._p(mof), C.c_int64(mof.shape[0]), _ffi._p(mdata),
        _ffi._p(moff), _ffi._p(pom), _ffi._p(s0), C.c_int64(len(m)), _ffi._p(pdata), _ffi._p(poff), C.c_int64(len(p)), _ffi._p(nof), _ffi._p(nd),
        _ffi._p(noff), C.c_int64(len(names)), _ffi._p(pof), _ffi._p(pd_), _ffi._p(paroff), C.c_int64(len(parents)),
        _ffi._p(order) if order.size else C.c_void_p(0), str(path).encode(), head.encode(), version_db.encode(),
        C.byref(n_lines)), "mirge_gff_write_device")
    tm["device_call_s"] = time.perf_counter() - t0
    return dict(records=None, rows=None, tables=tables, lines=int(n_lines.value), timing={k: round(v, 4) for k, v in tm.items()})


def write_gff(args, workDir, ref_db, base_names, casc: Cascade, uniq, res, seqs: FlatSeqs, ps, ref, counts, order):
    """``-gff``: ``sample_miRge3.gff``, the rows in the reference's order (exact-miRNA rows of the mapped frame, then
    the isomiR rows, :50-60) -- the host route: k_isotype's records fetched, the file built by ``mirge_gff_write``."""
    import time
    tm = {}
    t0 = time.perf_counter()
    tables = name_tables(args, ref_db, casc)
    tm["name_tables_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    po = ps[order]
    rows = np.concatenate([order[po == EXACT_PASS], order[po == ISO_PASS]]).astype(np.int64)
    recs = isomir_records(casc, uniq, res, tables, rows)
    tm["typing_call_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    version_db = "miRBase22" if ref_db == "miRBase" else "MirGeneDB2.0"
    head = ("# GFF3 adapted for miRNA sequencing data\n## VERSION 0.0.1\n## source-ontology: " + version_db + "\n## COLDATA: " +
            ",".join(str(nm) for nm in base_names) + "\n")
    # (the writer indexes the run's whole table of unique reads through `rows`: no gather of the rows' reads and counts here)
    names, parents = FlatSeqs.from_list(tables["printed"]), FlatSeqs.from_list(tables["parents"])
    nrow = np.ascontiguousarray(tables["name_of_ref"][ref[rows]], dtype=np.int32) if rows.size else np.zeros(1, np.int32)
    prow = np.ascontiguousarray(tables["parent_of_ref"][ref[rows]], dtype=np.int32) if rows.size else np.zeros(1, np.int32)
    cnt = np.ascontiguousarray(counts, dtype=np.uint32).reshape(len(seqs), len(base_names)) if len(seqs) else np.zeros((1, len(base_names)), np.uint32)
    sdata = np.ascontiguousarray(seqs.data) if seqs.data.size else np.zeros(1, np.uint8)
    soff = np.ascontiguousarray(seqs.offsets, dtype=np.int64)
    nd = np.ascontiguousarray(names.data) if names.data.size else np.zeros(1, np.uint8)
    pd_ = np.ascontiguousarray(parents.data) if parents.data.size else np.zeros(1, np.uint8)
    recs = np.ascontiguousarray(recs) if rows.size else np.zeros(1, dtype=RECORD)
    rows_c = np.ascontiguousarray(rows, dtype=np.int64) if rows.size else np.zeros(1, np.int64)
    tm["rows_of_the_file_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    _ffi._check(_ffi.load().mirge_gff_write(str(Path(workDir) / "sample_miRge3.gff").encode(), head.encode(), version_db.encode(),
                                            _ffi._p(recs), C.c_int64(rows.shape[0]), _ffi._p(sdata), _ffi._p(soff), _ffi._p(cnt),
                                            C.c_int32(len(base_names)), _ffi._p(nrow), _ffi._p(nd),
                                            _ffi._p(np.ascontiguousarray(names.offsets, dtype=np.int64)), C.c_int64(len(names)),
                                            _ffi._p(prow), _ffi._p(pd_), _ffi._p(np.ascontiguousarray(parents.offsets, dtype=np.int64)),
                                            C.c_int64(len(parents)), _ffi._p(rows_c), C.c_int64(len(seqs))), "mirge_gff_write")
    tm["write_s"] = time.perf_counter() - t0
    return dict(records=recs, rows=rows, tables=tables, timing={k: round(v, 4) for k, v in tm.items()})
