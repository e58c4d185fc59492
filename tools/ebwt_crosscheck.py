#!/usr/bin/env python3
"""Pin the .ebwt reader (mirge3_amd/ebwt.py) against indexes written by a REAL bowtie-build.

The reader has only ever read indexes written by tests/ebwt_writer.py -- the same author's reading of bowtie's ebwt.h.  Neither
this image nor the GPU pool holds a bowtie, so the row stays "unverified against a real index"; this tool closes the gap on any
machine that has one (no GPU needed):

  python tools/ebwt_crosscheck.py --bowtie-dir /path/with/bowtie-build [--fasta-dir DIR | --golden] [--large]

For every FASTA library (default: the golden cases' libraries under tests/golden, plus synthetic ones with N stretches, empty
and one-base references and long headers) it runs `bowtie-build [--large-index] <fa> <base>`, reads <base>.*.ebwt[l] with
ebwt.read_ebwt and compares names (text before the first blank, as the reference takes them: summary.py:776-790), full headers and
sequences with the FASTA -- and, when `bowtie-inspect` is in the same directory, with what `bowtie-inspect -n` and `bowtie-inspect`
print (the reference's own way to read an index).  Known, documented difference: bowtie-build drops empty references and
all-N references from the index; the report lists them instead of failing on them.
Exit code 0 = every library agrees.  The report is also written to gpurun_out/ebwt_crosscheck.txt.
"""
import argparse
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import mirge3_amd  # noqa: E402,F401
from mirge3_amd import ebwt  # noqa: E402
from mirge3_amd.seqio import read_fasta  # noqa: E402


def synthetic_fastas(d):
    rng = np.random.default_rng(12)

    def rs(n, pn=0.0):
        s = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=n))
        return "".join("N" if rng.random() < pn else c for c in s) if pn else s
    sets = {
        "plain": [(f"hsa-miR-{i}-5p MIMAT{i:07d} Homo sapiens", rs(int(rng.integers(18, 26)))) for i in range(300)],
        "ambiguous": [(f"ref{i} chr{i} segs:1-9 note", "NNNN" + rs(40) + "NN" + rs(7, 0.1) + "NNN") for i in range(40)] + [("one", "A"), ("tail", rs(64) + "N")],
        "long": [(f"mrna{i}", rs(int(rng.integers(500, 6000)), 0.001)) for i in range(60)],
    }
    out = []
    for name, recs in sets.items():
        p = os.path.join(d, name + ".fa")
        with open(p, "w") as fh:
            fh.write("".join(f">{h}\n{s}\n" for h, s in recs))
        out.append(p)
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--bowtie-dir", required=True)
    ap.add_argument("--fasta-dir", default=None, help="directory of *.fa libraries (default: tests/golden libraries + synthetic ones)")
    ap.add_argument("--large", action="store_true", help="also build every library with --large-index (.ebwtl, 64-bit offsets)")
    args = ap.parse_args(argv)
    build = os.path.join(args.bowtie_dir, "bowtie-build")
    inspect = os.path.join(args.bowtie_dir, "bowtie-inspect")
    if not os.path.exists(build):
        print(f"{build}: not found", file=sys.stderr)
        return 2
    tmp = tempfile.mkdtemp(prefix="mirge_ebwt_xcheck_")
    try:
        if args.fasta_dir:
            fastas = sorted(glob.glob(os.path.join(args.fasta_dir, "*.fa")))
        else:
            fastas = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "case[13]*", "libs", "*", "index.Libs", "*.fa"))) + synthetic_fastas(tmp)
        report, bad = [], 0

        def say(line):
            print(line)
            report.append(line)
        for fa in fastas:
            lib = read_fasta(fa)
            for large in ([False, True] if args.large else [False]):
                base = os.path.join(tmp, os.path.basename(fa)[:-3] + ("_l" if large else ""))
                r = subprocess.run([build, "-q"] + (["--large-index"] if large else []) + [fa, base], capture_output=True, text=True)
                if r.returncode != 0:
                    say(f"{fa}: bowtie-build failed: {r.stderr.strip()[-200:]}")
                    bad += 1
                    continue
                got = ebwt.read_ebwt(base)
                seqs = lib.seqs.to_list()
                # bowtie-build leaves out references without an unambiguous base
                kept = [i for i, s in enumerate(seqs) if any(c in "ACGTacgt" for c in s)]
                dropped = len(seqs) - len(kept)
                want_h = [lib.headers[i] for i in kept]
                want_s = [seqs[i].upper() for i in kept]
                ok_h = got.headers == want_h
                ok_n = got.names == [h.split()[0] if h.split() else "" for h in want_h]
                ok_s = got.seqs.to_list() == want_s
                line = f"{os.path.basename(fa):32s} {'ebwtl' if large else 'ebwt '} refs {len(kept):6d} (dropped by bowtie-build: {dropped})  headers {'ok' if ok_h else 'DIFFER'}  names {'ok' if ok_n else 'DIFFER'}  sequences {'ok' if ok_s else 'DIFFER'}"
                if os.path.exists(inspect):
                    names = subprocess.run([inspect, "-n", base], capture_output=True, text=True).stdout.split("\n")
                    names = names[:-1] if names and names[-1] == "" else names
                    ok_i = names == got.headers
                    line += f"  bowtie-inspect -n {'ok' if ok_i else 'DIFFERS'}"
                    ok_h = ok_h and ok_i
                say(line)
                if not (ok_h and ok_n and ok_s):
                    bad += 1
                    for k, (a, b) in enumerate(zip(got.seqs.to_list(), want_s)):
                        if a != b:
                            say(f"     first differing sequence: reference {k} ({want_h[k][:40]}): read {a[:50]}... FASTA {b[:50]}...")
                            break
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    say("every index read back as its FASTA" if bad == 0 else f"{bad} libraries differ")
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "ebwt_crosscheck.txt"), "w") as fh:
            fh.write(f"bowtie-build: {build}\n" + "\n".join(report) + "\n")
    except OSError:
        pass
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
