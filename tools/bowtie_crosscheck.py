#!/usr/bin/env python3
"""Pin the alignment predicate against a REAL bowtie 1.x wherever one is installed (SURVEY.md 8c "live cross-check").

The cascade's arithmetic is third-party bowtie; this image and the GPU pool have none, hence "parity unpinned" for
the predicate.  This tool closes that gap on any machine that has bowtie + an MI355X:

  python tools/bowtie_crosscheck.py --bowtie-dir /path/with/bowtie [--libs <dir>/<org>/index.Libs --org human --db miRBase]
                                    [--reads reads.txt | --synthetic 20000]

For every pass it writes the FASTA the reference would write (name = sequence; T-tail stripped for pass 3), runs
`bowtie <index> <the reference's argument string> <threads> <fasta>` verbatim (manifoldAlign.py:85), parses the SAM as
the reference does (field 0 / field 2) and compares with ONE pass of the GPU engine on the same reads:
  * membership (aligned / not aligned) must be identical -- this is what the per-class counts depend on;
  * the reference named may differ where several hits are equally good (reported, not an error): `other-name` per
    pass and, at the end, the TIE RATE -- the share of aligned reads whose reported name differs, i.e. how often the
    documented tie-break (fewest mismatches, lowest reference, leftmost offset) decides something bowtie decides otherwise;
  * the whole report is also written to gpurun_out/bowtie_crosscheck.txt (scratch, merged back by gpurun).
Indexes: `<index>.1.ebwt` is built with `bowtie-build` from `<index>.fa` if missing (skipped when the directory has
no bowtie-build, e.g. the shim in mirge3.0_amd/shim, which answers from `<index>.fa`).
Exit code 0 = every pass agrees on membership.

A read on which the two disagree is printed with the open question(s) of the restatement it falls under (DESIGN.md section 3,
SURVEY.md 8 caveats i-v, and vi of round 4's review), so that an `only-bowtie` / `only-gpu` line of the first real run explains
itself:
  (i)   -n mode, read longer than the 28-nt seed: Maq rounding of Q40 -> 30 with -e 70 => at most 2 mismatches overall, the
        second only beyond the seed
  (ii)  the read holds an N (scored as a mismatch at every alignment here)
  (iii) what bowtie is handed is empty or no longer than the mismatch budget (pass 3 heads, -5/-3 trims): skipped here
  (iv)  the library holds ambiguous bases: windows over them are no hits here (bowtie excludes them from its index)
  (v)   several equally good hits: the NAME may differ (never membership) -- counted as `other-name`
  (vi)  --maxbts, bowtie's backtracking limit ("may cause some valid alignments to be missed"), is documented for -n 2 / -n 3
        only; the cascade's strings are -n 0 / -n 1 / -v: none of them is under it, so it explains NO line of this report
"""
import argparse
import os
import re
import shlex
import subprocess
import sys
import tempfile

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import mirge3_amd  # noqa: E402,F401
from mirge3_amd import _ffi, synth  # noqa: E402
from mirge3_amd.cascade import PASSES, policies  # noqa: E402
from mirge3_amd.seqio import FlatSeqs, index_basename, load_library_dir, write_fasta  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--bowtie-dir", required=True)
    ap.add_argument("--libs", default=None, help="<lib>/<org>/index.Libs with <index>.fa files; default: synthetic 'ci' libraries")
    ap.add_argument("--org", default="human")
    ap.add_argument("--db", default="miRBase")
    ap.add_argument("--reads", default=None, help="one sequence per line; default: synthetic reads")
    ap.add_argument("--synthetic", type=int, default=20000)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--threads", type=int, default=4)
    args = ap.parse_args(argv)
    bowtie = os.path.join(args.bowtie_dir, "bowtie")
    build = os.path.join(args.bowtie_dir, "bowtie-build")
    tmp = tempfile.mkdtemp(prefix="mirge_xcheck_")
    if args.libs:
        idxdir = args.libs
        libs = load_library_dir(os.path.dirname(os.path.dirname(idxdir)), args.org, args.db)
        sl = None
    else:
        sl = synth.make_libraries(seed=5, scale="ci")
        libs = sl.libs
        idxdir = os.path.join(tmp, "index.Libs")
        os.makedirs(idxdir)
        for key, lib in libs.items():
            write_fasta(os.path.join(idxdir, index_basename(args.org, key, args.db) + ".fa"), lib)
    if args.reads:
        seqs = [ln.strip() for ln in open(args.reads) if ln.strip()]
    else:
        sl = sl or synth.make_libraries(seed=5, scale="ci")
        seqs = sorted(set(synth.make_reads(sl, args.synthetic, seed=9, n_frac=0.01).to_list()))
    if os.path.exists(build):
        for key in libs:
            base = os.path.join(idxdir, index_basename(args.org, key, args.db))
            if not os.path.exists(base + ".1.ebwt"):
                subprocess.run([build, "-q", base + ".fa", base], check=True, stdout=subprocess.DEVNULL)
    ctx = _ffi.Context(args.device)
    dev = {k: _ffi.DeviceLibrary(ctx, v.seqs) for k, v in libs.items()}
    pol = policies(9)
    annotated = set()
    bad = 0
    n_both = n_other = 0
    report = []

    def say(line):
        print(line)
        report.append(line)
    lib_has_n = {k: bool(np.isin(v.seqs.data, np.frombuffer(b"ACGTacgtUu", np.uint8), invert=True).any()) for k, v in libs.items()}

    def caveats(it, key, x):
        """the open questions (i)-(iv) a read handed to pass `it` as `x` falls under"""
        kw = PASSES[it][3]
        out = []
        eff = len(x) - kw.get("trim5", 0) - kw.get("trim3", 0)
        if kw["mode"] == 0 and eff > kw["seedlen"]:
            out.append("(i) longer than the seed in -n mode")
        if "N" in x.upper():
            out.append("(ii) N in the read")
        if eff < 1 or eff <= kw["mm"]:
            out.append("(iii) empty or no longer than the mismatch budget")
        if lib_has_n[key]:
            out.append("(iv) the library holds ambiguous bases")
        return out or ["none of (i)-(iv): a difference in the predicate itself"]

    fasta = os.path.join(tmp, "bwtInput.fasta")
    for it in range(9):
        col, key, argstr, _ = PASSES[it]
        if key not in libs:
            continue
        if it == 0:
            recs = [(q, q) for q in seqs if len(q) < 26]
        elif it == 1:
            recs = [(q, q) for q in seqs if len(q) > 25]
        else:
            un = [q for q in seqs if q not in annotated]
            recs = [(q, q[:re.search('T{3,}$', q).start()]) for q in un if re.search('T{3,}$', q)] if it == 3 else [(q, q) for q in un]
        with open(fasta, "w") as fh:
            fh.write("".join(f">{q}\n{x}\n" for q, x in recs))
        cmd = [bowtie, os.path.join(idxdir, index_basename(args.org, key, args.db))] + shlex.split(argstr) + [str(args.threads), fasta]
        out = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True).stdout
        ref_hit = {}
        for ln in out.split("\n"):
            if ln and not ln.startswith("@"):
                f = ln.split("\t")
                if f[2] != "*":
                    ref_hit[f[0]] = f[2]  # last record wins, as in manifoldAlign.py:50-56
        # the same reads through ONE pass of the GPU engine (the pass's own subset rule and T-tail strip apply)
        names = [q for q, _ in recs]
        got = {}
        if names:
            dr = _ffi.DeviceReads.pack(ctx, FlatSeqs.from_list(names))
            res = _ffi.cascade_run(ctx, dr, [dev[key]], [pol[it]])
            ps, ref, _, _ = res.fetch()
            got = {q: libs[key].names[int(r)] for q, p, r in zip(names, ps, ref) if p == 0}
            res.close(); dr.close()
        only_b = sorted(set(ref_hit) - set(got))
        only_g = sorted(set(got) - set(ref_hit))
        diff_name = sum(1 for q in got if q in ref_hit and ref_hit[q] != got[q])
        say(f"pass {it} {col:14s} reads {len(recs):7d}  bowtie {len(ref_hit):7d}  gpu {len(got):7d}  only-bowtie {len(only_b)}  "
            f"only-gpu {len(only_g)}  other-name {diff_name}")
        handed = dict(recs)
        tally = {}
        for q in only_b + only_g:
            for c in caveats(it, key, handed[q]):
                tally[c] = tally.get(c, 0) + 1
        for c, k in sorted(tally.items()):
            say(f"     {k:6d} of the differing reads: {c}")
        for q in (only_b + only_g)[:5]:
            say(f"     {q} bowtie: {ref_hit.get(q)} gpu: {got.get(q)}  [{'; '.join(caveats(it, key, handed[q]))}]")
        for q in [q for q in got if q in ref_hit and ref_hit[q] != got[q]][:3]:
            say(f"     name differs: {q} bowtie: {ref_hit[q]} gpu: {got[q]}")
        n_both += sum(1 for q in got if q in ref_hit)
        n_other += diff_name
        bad += len(only_b) + len(only_g)
        annotated |= set(ref_hit)  # the cascade continues with what the REFERENCE path annotated
    say("membership identical in every pass" if bad == 0 else f"{bad} reads differ in membership")
    say("(vi) --maxbts applies to -n 2 / -n 3 only: no argument string of the cascade is under it")
    say(f"tie rate: {n_other} of {n_both} reads aligned by both carry another reference name ({100.0 * n_other / max(n_both, 1):.3f} %)")
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bowtie_crosscheck.txt"), "w") as fh:
            fh.write(f"bowtie: {bowtie}\n" + "\n".join(report) + "\n")
    except OSError:
        pass
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
