#!/usr/bin/env python3
"""Pin the trimming restatement against a REAL cutadapt wherever one is installed (twin of tools/bowtie_crosscheck.py).

Row N4's modifiers (quality / NextSeq trimming, 3' and 5' adapters, N ends, cuts) are cutadapt's; this image has no
cutadapt, so `oracle.trim_stages` -- and `k_trim`, which equals it -- are restated from cutadapt's published algorithms
("parity unpinned").  On a machine with cutadapt on PATH:

  python tools/cutadapt_crosscheck.py [--records 6000] [--gpu]

For each of fourteen option sets (the chains of tests/test_gpu_parity.py's trimming tests: quality / NextSeq trimming, 3' and 5'
adapters, two adapters, -n, --no-indels, wildcards) it writes a FASTQ file of synthetic records (adapters whole, partial,
with substitutions / indels, low-quality tails, N ends), runs `cutadapt` with the equivalent command line, and compares
the trimmed sequence of EVERY record with the oracle's last stage -- and, with --gpu, with mirge_reads_parse_trim.
Without cutadapt it says so and exits 0 (skip).  The report goes to stdout and gpurun_out/cutadapt_crosscheck.txt.
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

A3 = "TGGAATTCTCGGGTGCCAAGGAACTCCAG"
A5 = "GTTCAGAGTTCTACAGTCCGACGATC"
OPTION_SETS = [
    dict(q_back=10),
    dict(q_back=10, adapter=A3),
    dict(q_back=20, q_front=8, adapter=A3, trim_n=True, cut=[2, -1]),
    dict(q_back=10, adapter="AGATCGGAAGAGCNNNNACGT", error_rate=0.2, overlap=5, nextseq=20),
    dict(adapter=A3, cut=[-3]),
    dict(q_back=10, adapter=A5, front=True),
    dict(q_back=10, adapter=A5, front=True, overlap=5, trim_n=True),
    dict(q_back=10, adapters=[("back", A3), ("front", A5)]),
    dict(q_back=10, adapters=[("front", A5), ("back", A3)]),
    dict(q_back=15, adapter=A3, error_rate=0.05),
    dict(q_back=10, adapter=A3, times=3),
    dict(q_back=10, adapter=A3, indels=False),
    dict(q_back=10, adapter=A3, read_wildcards=True),
    dict(q_back=10, adapter="TGGAATTCNNGGGTGCCAAGGAACTCCAG", adapter_wildcards=False),
]


def records(rng, n, ads):
    out = []
    for i in range(n):
        ins = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=int(rng.integers(12, 41))))
        parts = {}
        for kind, ad in ads:
            x = list(ad.replace("N", "A"))
            r = rng.random()
            if r < 0.15:
                x[int(rng.integers(0, len(x)))] = "ACGT"[int(rng.integers(0, 4))]
            elif r < 0.22:
                del x[int(rng.integers(1, len(x) - 1))]
            elif r < 0.29:
                x.insert(int(rng.integers(1, len(x) - 1)), "ACGT"[int(rng.integers(0, 4))])
            elif r < 0.45:
                x = x[int(rng.integers(3, len(x) - 2)):] if kind == "front" else x[:int(rng.integers(1, len(x) - 2))]
            elif r < 0.55:
                x = []
            parts[kind] = parts.get(kind, "") + "".join(x)
        seq = (parts.get("front", "") + ins + parts.get("back", "") + "ACGTTGCA"[: int(rng.integers(0, 9))])[: int(rng.integers(36, 101))]
        if rng.random() < 0.05:
            seq = "N" * int(rng.integers(1, 3)) + seq[2:-1] + "N"
        q = np.full(len(seq), ord("I"), dtype=np.uint8)
        if rng.random() < 0.3:
            k = int(rng.integers(1, 15))
            q[-k:] = rng.integers(33, 50, size=min(k, len(seq)))
        if rng.random() < 0.1:
            q[: int(rng.integers(1, 5))] = 35
        out.append((seq, q.tobytes().decode()))
    return out


def cutadapt_argv(o, src, dst):
    a = ["cutadapt", "-j", "1", "-o", dst]
    if o.get("nextseq") is not None:
        a += ["--nextseq-trim", str(o["nextseq"])]
    if o.get("q_back") is not None:
        a += ["-q", f"{o.get('q_front', 0)},{o['q_back']}"]
    for kind, ad in o.get("adapters", [("front" if o.get("front") else "back", o["adapter"])] if o.get("adapter") else []):
        a += ["-g" if kind == "front" else "-a", ad]
    a += ["-e", str(o.get("error_rate", 0.12)), "-O", str(o.get("overlap", 3))]
    if o.get("times", 1) > 1:
        a += ["-n", str(o["times"])]
    if not o.get("indels", True):
        a += ["--no-indels"]
    if o.get("read_wildcards"):
        a += ["--match-read-wildcards"]
    if not o.get("adapter_wildcards", True):
        a += ["-N"]
    if o.get("trim_n"):
        a += ["--trim-n"]
    for c in o.get("cut", []):
        a += ["-u", str(c)]
    return a + [src]


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=6000)
    ap.add_argument("--gpu", action="store_true")
    args = ap.parse_args(argv)
    lines = []

    def say(s):
        print(s)
        lines.append(s)

    exe = shutil.which("cutadapt")
    if exe is None:
        say("cutadapt_crosscheck: SKIPPED -- no `cutadapt` on PATH (the trimming restatement stays unpinned on this machine)")
        rc = 0
    else:
        import oracle
        say("cutadapt: " + subprocess.run([exe, "--version"], capture_output=True, text=True).stdout.strip())
        tmp = tempfile.mkdtemp(prefix="mirge_cutadapt_")
        rc = 0
        for k, o in enumerate(OPTION_SETS):
            ads = o.get("adapters", [("front" if o.get("front") else "back", o["adapter"])] if o.get("adapter") else [("back", A3)])
            recs = records(np.random.default_rng(100 + k), args.records, ads)
            src, dst = os.path.join(tmp, f"in{k}.fastq"), os.path.join(tmp, f"out{k}.fastq")
            with open(src, "w") as fh:
                fh.write("".join(f"@r{i}\n{s}\n+\n{q}\n" for i, (s, q) in enumerate(recs)))
            subprocess.run(cutadapt_argv(o, src, dst), check=True, capture_output=True, text=True)
            with open(dst) as fh:
                got = [ln.rstrip("\n") for ln in fh][1::4]
            oo = dict(o)
            oo.setdefault("q_back", None)
            want = [(oracle.trim_stages(s, q, oo) or [s])[-1] for s, q in recs]
            bad = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
            say(f"option set {k}: {o} -> {len(recs)} records, {len(bad)} differ from the oracle")
            for i in bad[:5]:
                say(f"    {recs[i][0]} cutadapt: {got[i]} oracle: {want[i]}")
            rc |= 1 if bad or len(got) != len(want) else 0
            if args.gpu:
                import mirge3_amd  # noqa: F401
                from mirge3_amd import _ffi
                ctx = _ffi.Context(0)
                a1 = ads[0] if o.get("adapter") or o.get("adapters") else (None, None)
                a2 = ads[1] if len(ads) > 1 else (None, None)
                trim = _ffi.MirgeTrim.make(adapter=a1[1], front=a1[0] == "front", adapter2=a2[1], front2=a2[0] == "front",
                                           quality_back=o.get("q_back", -1) if o.get("q_back") is not None else -1,
                                           quality_front=o.get("q_front", 0), nextseq=o.get("nextseq", -1) if o.get("nextseq") is not None else -1,
                                           min_overlap=o.get("overlap", 3), error_rate=o.get("error_rate", 0.12),
                                           trim_n=o.get("trim_n", False), cut=o.get("cut", []), count_per_modifier=False,
                                           times=o.get("times", 1), indels=o.get("indels", True), read_wildcards=o.get("read_wildcards", False),
                                           adapter_wildcards=o.get("adapter_wildcards", True))
                raw, _ = _ffi.DeviceReads.parse(ctx, open(src, "rb").read(), 1, 0, trim)
                gpu = raw.unpack().to_list()
                raw.close()
                badg = sum(1 for a, b in zip(gpu, got) if a.upper() != b.upper().replace("U", "T"))
                say(f"        k_trim vs cutadapt: {badg} of {len(got)} differ")
                rc |= 1 if badg else 0
        shutil.rmtree(tmp, ignore_errors=True)
        say("trimming restatement == cutadapt on every record" if rc == 0 else "DIFFERENCES found: see above")
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "cutadapt_crosscheck.txt"), "w") as fh:
            fh.write("\n".join(lines) + "\n")
    except OSError:
        pass
    return rc


if __name__ == "__main__":
    sys.exit(main())
