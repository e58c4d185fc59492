"""Annotation cascade -- host mirror of ``mirge/libs/manifoldAlign.py``.

``bwt_align(args, pdDataFrame, workDir, ref_db)`` has the reference's signature and returns the
same DataFrame (``annotFlag`` int, the nine/ten annotation columns as ``str``, sample columns),
so the reference's ``summarize`` could consume it.  The ten bowtie runs, their FASTA temp files
and the SAM parsing (``alignPlusParse``, manifoldAlign.py:12-64) are replaced by ONE call into
the HIP kernels: ``mirge_cascade_run``.
"""
from __future__ import annotations

import os
import shlex
import time
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import PASS_COLUMNS, _ffi
from .seqio import FlatSeqs, Library, load_library_dir

# (column written, library key, bowtie argument string it stands for, policy)
# manifoldAlign.py:84-85 verbatim; predicate per SURVEY.md 8 table a8-P.
PASSES = [
    ("exact miRNA",   "mirna",        " -n 0 -f --norc -S --threads ",
     dict(mode=0, mm=0, seedlen=28, maxtotal=2, len_lt=26)),                 # :93  len < 26
    ("hairpin miRNA", "hairpin",      " -n 1 -f --norc -S --threads ",
     dict(mode=0, mm=1, seedlen=28, maxtotal=2, len_gt=25)),                 # :104 len > 25
    ("mature tRNA",   "mature_trna",  " -v 1 -f -a --best --strata --norc -S --threads ",
     dict(mode=1, mm=1, seedlen=28, maxtotal=1)),
    ("primary tRNA",  "pre_trna",     " -v 0 -f -a --best --strata --norc -S --threads ",
     dict(mode=1, mm=0, seedlen=28, maxtotal=0, ttail=1)),                   # :118-126
    ("snoRNA",        "snorna",       " -n 1 -f --norc -S --threads ",
     dict(mode=0, mm=1, seedlen=28, maxtotal=2)),
    ("rRNA",          "rrna",         " -n 1 -f --norc -S --threads ",
     dict(mode=0, mm=1, seedlen=28, maxtotal=2)),
    ("ncrna others",  "ncrna_others", " -n 1 -f --norc -S --threads ",
     dict(mode=0, mm=1, seedlen=28, maxtotal=2)),
    ("mRNA",          "mrna",         " -n 0 -f --norc -S --threads ",
     dict(mode=0, mm=0, seedlen=28, maxtotal=2)),
    ("isomiR miRNA",  "mirna",        " -5 1 -3 2 -v 2 -f --norc --best -S --threads ",
     dict(mode=1, mm=2, seedlen=28, maxtotal=2, trim5=1, trim3=2)),
    ("spike-in",      "spike-in",     " -n 0 -f --norc -S --threads ",
     dict(mode=0, mm=0, seedlen=28, maxtotal=2)),
]
EXACT_PASS, ISO_PASS = 0, 8
assert [p[0] for p in PASSES] == PASS_COLUMNS


def policies(n_pass: int = 9) -> List[_ffi.MirgePolicy]:
    out = []
    for _, _, _, kw in PASSES[:n_pass]:
        p = _ffi.MirgePolicy()
        for k, v in kw.items():
            setattr(p, k, v)
        out.append(p)
    return out


class Cascade:
    """The libraries of one organism resident on one GPU + the pass table."""

    def __init__(self, ctx: _ffi.Context, libs: Dict[str, Library], spike_in: bool = False,
                 n_pass: Optional[int] = None):
        self.ctx = ctx
        self.libs = libs
        self.n_pass = n_pass if n_pass is not None else (10 if spike_in else 9)
        self._dev: Dict[str, _ffi.DeviceLibrary] = {}
        self.dev_libs: List[Optional[_ffi.DeviceLibrary]] = []
        self.timing: Dict[str, float] = {}  # where a process's first sample spends its library time (bench.py cli_path)
        for _, key, _, _ in PASSES[:self.n_pass]:
            if key not in libs:
                self.dev_libs.append(None)
                continue
            if key not in self._dev:  # the miRNA library serves passes 0 and 8
                t0 = time.perf_counter()
                self._dev[key] = _ffi.DeviceLibrary(ctx, libs[key].seqs)
                self.timing["pack_upload_s"] = self.timing.get("pack_upload_s", 0.0) + time.perf_counter() - t0
                target = getattr(libs[key], "cache_target", None)
                if target is not None:  # read from its FASTA / .ebwt just now: keep the packed image next to the index
                    from . import libcache
                    t0 = time.perf_counter()
                    try:  # the cache is optional: whatever goes wrong while writing it must not end the run
                        libcache.save(target[0], libs[key], self._dev[key].packed_image(), target[1])
                    except Exception as e:  # noqa: BLE001
                        import logging
                        logging.getLogger("mirge3_amd").warning("library cache of %s not written: %r", target[0], e)
                    libs[key].cache_target = None
                    self.timing["cache_write_s"] = self.timing.get("cache_write_s", 0.0) + time.perf_counter() - t0
            self.dev_libs.append(self._dev[key])
        self.policies = policies(self.n_pass)
        self._prepared = _ffi.cascade_args(self.dev_libs, self.policies)

    def lib_of_pass(self, p: int) -> Library:
        return self.libs[PASSES[p][1]]

    def run(self, reads: _ffi.DeviceReads) -> _ffi.CascadeResult:
        return _ffi.cascade_run(self.ctx, reads, self.dev_libs, self.policies, self._prepared)

    def prepare(self, reads: _ffi.DeviceReads):
        """The tables a cascade over ``reads`` builds on first use, built now (``mirge_cascade_prepare``); a no-op once they
        exist for these libraries, policies and read lengths."""
        _ffi.cascade_prepare(self.ctx, reads, self.dev_libs, self.policies, self._prepared)

    def collapse_and_run(self, raw: _ffi.DeviceReads):
        """One sample's raw reads -> (unique reads, annotation): collapse and cascade as one call, the bulk group's
        passes queued behind the collapse kernels without waiting for the host (``mirge_collapse_cascade``)."""
        return _ffi.collapse_cascade(self.ctx, raw, self.dev_libs, self.policies, self._prepared)

    def annotate(self, seqs: FlatSeqs):
        """Convenience: host sequences in, (pass, ref, off, mm) numpy arrays out."""
        dr = _ffi.DeviceReads.pack(self.ctx, seqs)
        res = self.run(dr)
        out = res.fetch()
        res.close()
        dr.close()
        return out

    def close(self):
        for d in self._dev.values():
            d.close()
        self._dev.clear()


_cascade_cache: Dict[tuple, Cascade] = {}


def get_cascade(args, ref_db: str, device: int = 0) -> Cascade:
    key = (os.path.abspath(str(args.libraries_path)), args.organism_name, ref_db, bool(args.spikeIn), device)
    if key not in _cascade_cache:
        t0 = time.perf_counter()
        libs = load_library_dir(str(args.libraries_path), args.organism_name, ref_db, with_spike=bool(args.spikeIn))
        t_read = time.perf_counter() - t0
        ctx = _ffi.Context(device)
        t_ctx = time.perf_counter() - t0 - t_read
        _cascade_cache[key] = Cascade(ctx, libs, spike_in=bool(args.spikeIn))
        _cascade_cache[key].timing.update(read_index_or_cache_s=t_read, context_s=t_ctx)
    return _cascade_cache[key]


def bwt_align(args, pdDataFrame, workDir, ref_db):
    """Drop-in for ``bwtAlign`` (manifoldAlign.py:68-146)."""
    begningTime = time.perf_counter()
    runlogFile = Path(workDir) / "run.log"
    outlog = open(str(runlogFile), "a+")
    if not args.quiet:
        print("Alignment in progress ...")
    outlog.write("Alignment in progress ...\n")
    casc = get_cascade(args, ref_db, getattr(args, "device", 0))
    seqs = FlatSeqs.from_list([str(s) for s in pdDataFrame.index])
    ps, ref, off, mm = casc.annotate(seqs)
    colnames = list(pdDataFrame.columns)
    flag = pdDataFrame[colnames[0]].to_numpy().copy()
    for p in range(casc.n_pass):
        sel = np.nonzero(ps == p)[0]
        if sel.size == 0:
            continue
        names = np.asarray(casc.lib_of_pass(p).names, dtype=object)
        col = pdDataFrame[colnames[1 + p]].to_numpy(dtype=object).copy()
        col[sel] = names[ref[sel]]
        pdDataFrame[colnames[1 + p]] = col
        flag[sel] = 1
    pdDataFrame[colnames[0]] = flag
    finish = time.perf_counter()
    if not args.spikeIn:
        pdDataFrame = pdDataFrame.drop(columns=['spike-in'])
    pdDataFrame = pdDataFrame.fillna('')
    if not args.quiet:
        print(f'Alignment completed in {round(finish-begningTime, 4)} second(s)\n')
    outlog.write(f'Alignment completed in {round(finish-begningTime, 4)} second(s)\n')
    outlog.close()
    return pdDataFrame


def bwt_align_bowtie(args, pdDataFrame, workDir, ref_db):
    """``--backend bowtie``: the reference's own cascade across its process boundary -- what ``bwtAlign`` / ``alignPlusParse`` do
    (manifoldAlign.py:12-64,68-146): per pass the subset rule, a FASTA whose record names are the sequences (the head without
    its T run for pass 3), ``<bowtie_path>/bowtie <index><argument string verbatim><threads> <fasta>``, every SAM line with a
    reference name written into the pass's column (the last line of a read wins).  BASELINE.md's C1 through this build's CLI,
    and a user-side parity switch: the same FASTQ through ``--backend gpu`` and ``--backend bowtie`` must give the same tables
    wherever a real bowtie 1.x is installed (none in this image or on the GPU pool: tests run it against the stand-ins).
    Indexes: ``<index>.1.ebwt`` as the reference expects; built with ``bowtie-build`` from ``<index>.fa`` when missing."""
    import re
    import shutil
    import subprocess
    begningTime = time.perf_counter()
    outlog = open(str(Path(workDir) / "run.log"), "a+")
    if not args.quiet:
        print("Alignment in progress (bowtie backend) ...")
    outlog.write("Alignment in progress (bowtie backend) ...\n")
    bdir = getattr(args, "bowtie_path", None)
    bowtie = str(Path(bdir) / "bowtie") if bdir else (shutil.which("bowtie") or "bowtie")
    build = str(Path(bdir) / "bowtie-build") if bdir else shutil.which("bowtie-build")
    threads = int(getattr(args, "threads", 0) or 0) or (os.cpu_count() or 1)
    indexPath = Path(args.libraries_path) / args.organism_name / "index.Libs"
    # manifoldAlign.py:84-85: index name fragments and argument strings, in pass order
    indexNames = ['_mirna_', '_hairpin_', '_mature_trna', '_pre_trna', '_snorna', '_rrna', '_ncrna_others', '_mrna', '_mirna_', '_spike-in']
    colnames = list(pdDataFrame.columns)
    n_iter = 10 if args.spikeIn else 9
    fasta = Path(workDir) / "bwtInput.fasta"
    for it in range(n_iter):
        name = indexNames[it] + ref_db if it in (0, 1, 8) else indexNames[it]
        base = indexPath / (args.organism_name + name)
        if not Path(str(base) + ".1.ebwt").exists() and not Path(str(base) + ".1.ebwtl").exists():
            if Path(str(base) + ".fa").exists() and build and Path(build).exists():
                subprocess.run([build, "-q", str(base) + ".fa", str(base)], check=True, stdout=subprocess.DEVNULL)
            elif not Path(str(base) + ".fa").exists():  # (a stand-in bowtie may answer from <index>.fa; a real one says what it misses)
                # the reference's bwtAlign dies here with bowtie's CalledProcessError (manifoldAlign.py:19): a class silently
                # missing from the count tables is the one thing a parity switch must not do.  Only the optional spike-in
                # library (manifoldAlign.py:86-89) may be absent.
                if it == 9:
                    outlog.write(f"WARNING: no spike-in index {base}.1.ebwt: pass skipped\n")
                    continue
                outlog.close()
                raise FileNotFoundError(f"--backend bowtie: pass {it} ({colnames[1 + it]}): neither {base}.1.ebwt nor {base}.fa exists")
        seqs = pdDataFrame.index
        if it == 0:
            recs = [(q, q) for q in seqs if len(q) < 26]                      # :93
        elif it == 1:
            recs = [(q, q) for q in seqs if len(q) > 25]                      # :104
        else:
            un = pdDataFrame.index[pdDataFrame[colnames[0]].eq(0)]            # :120,129
            if it == 3:                                                       # :118-126
                recs = [(q, q[:re.search('T{3,}$', q).start()]) for q in un if re.search('T{3,}$', q)]
            else:
                recs = [(q, q) for q in un]
        with open(fasta, "w") as fh:
            fh.write("".join(f">{q}\n{x}\n" for q, x in recs))
        # (the reference joins these into one string for a shell, manifoldAlign.py:19,97-99; the same words as an argument
        #  list, so that a library path with a blank or a quote in it is a path and nothing else)
        cmd = [bowtie, str(base)] + shlex.split(PASSES[it][2]) + [str(threads), str(fasta)]
        sam = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True).stdout
        hit = {}
        for ln in sam.split("\n"):                                            # :50-56
            if ln and not ln.startswith("@"):
                f = ln.split("\t")
                if len(f) > 2 and f[2] != "*":
                    hit[f[0]] = f[2]
        if hit:
            idx = pdDataFrame.index.get_indexer(list(hit))
            if (idx < 0).any():  # a QNAME that is no row of the frame (the reference's `df.at[...] = ...` would silently grow the frame by a row)
                raise KeyError(f"bowtie reported a read that was not in its input: {list(hit)[int(np.argmin(idx))]!r}")
            col = pdDataFrame[colnames[1 + it]].to_numpy(dtype=object).copy()
            col[idx] = list(hit.values())
            pdDataFrame[colnames[1 + it]] = col
            flag = pdDataFrame[colnames[0]].to_numpy().copy()
            flag[idx] = 1
            pdDataFrame[colnames[0]] = flag
    try:
        os.remove(fasta)
    except OSError:
        pass
    finish = time.perf_counter()
    if not args.spikeIn:
        pdDataFrame = pdDataFrame.drop(columns=['spike-in'])
    pdDataFrame = pdDataFrame.fillna('')
    if not args.quiet:
        print(f'Alignment completed in {round(finish-begningTime, 4)} second(s)\n')
    outlog.write(f'Alignment completed in {round(finish-begningTime, 4)} second(s)\n')
    outlog.close()
    return pdDataFrame
