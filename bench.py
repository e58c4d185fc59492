#!/usr/bin/env python3
"""bench.py -- the hot path on synthetic data: collapse -> annotation cascade -> count join.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A step is one pass of the whole hot path over one sample whose packed reads are already resident
in HBM: mirge_collapse -> mirge_cascade_run (9 passes) -> mirge_count_join (count tables back on
the host).  Default workload = BASELINE.json configs[2] ("C3"): a 10 M-read human-like sample
against human-sized libraries.  `--workload c2` is configs[1] (pass 0 only).
One rank per GPU, one sample per rank, no collective on the data path ("scaling": "weak");
torch.distributed is used only for the barrier and the max-over-ranks time.

Rank 0 prints ONE JSON line with `roofline` (dominant kernel, HIP-event timed inside the timed
region) and, at N = 1, `cpu_baseline` (the oracle on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

RANDOM_SECTOR_PEAK_G = 55.0  # measured, tools/microbench_random.hip: L2-missing random loads, G/s
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
N_CU, SIMD_PER_CU, VALU_CYCLES_PER_WAVE64 = 256, 4, 4  # same guide: 256 CUs x 4 SIMD16; a wave64 vector instruction issues over 4 cycles
SCLK_PEAK_MHZ = 2400.0

# SURVEY.md 8(d), "Cascade (fused minimum)": 9 B in (8 B packed read + 1 B length) + 5 B out (1 B pass + 4 B reference id)
# per COLLAPSED read, + 8 B per further word of a wider read.  The unit of k_cascade_bulk / k_cascade_fused -- one launch for
# all passes of a read group -- is therefore the read of the group, not the read handed to a pass.
CASCADE_BYTES_PER_READ = 9 + 5

# Algorithmic bytes per unit (DESIGN.md "Kernels"): a packed short read is
# 8 B + 1 B length; an annotation is pass 1 B + position 4 B (+ mismatches 1 B, not counted).
ALGO_BYTES = {
    "k_collapse_insert": 9 + 4,      # read in, slot id out, per raw read
    "k_collapse_insert_key": 9 + 4,  # same, 64-bit-key table (<=31 nt, no N, one sample)
    "k_part_agg": 9 + 16, "k_part_split": 16 + 16, "k_part_dedup": 16 + 17,  # 17 = key + len + count + first per unique (bound: per raw read)
    "k_heads_blocksum": 4 + 4 + 1,   # slot id + first-index in, head flag out, per raw read
    "k_collapse_scatter": 1 + 4 + 13,  # head flag + slot id in; key+len+count out (upper bound: per raw read)
    "k_pass": 4 + 9 + 5,             # active index + read in, annotation or survivor index out, per read handed to the pass
    "k_cascade_fused": 9 + 6 + 8,   # small groups, whole cascade in one launch: read in; pass+position+mismatches, ref+offset out
    "k_resolve": 5 + 8,              # pass+position in, ref+offset out
    "k_join": 5 + 4,                 # pass+ref + one count in (S = 1)
    "k_member_list": 5 + 4 + 4,      # pass+ref (+ count or slot) in, member index out, per read of the group
    "k_tally": 5 + 9 + 4,            # pass+ref, read, one count in per read of the group (S = 1); the tables are a few MB per launch
    "k_isotype": 5 + 9 + 336,        # pass+ref, read in; one typed record out per miRNA read
    "k_trim": 2 * 50 + 16,           # sequence + quality bytes of a 50-cycle record in, new bounds out
    "k_scan_blocksums": 8,
}


def algo_bytes(name, per_pass=False):
    """per-unit bytes of a profile record such as 'k_pass[6].w1' (w1/w2/w4 = 1/2/4-word reads).  The whole-cascade kernels
    (k_cascade_bulk, k_cascade_fused) are priced per collapsed read (14 B); per_pass=True gives k_pass's 18 B per
    (read, pass) for the bulk kernel's secondary figure."""
    base, _, w = name.partition(".w")
    w = w.rstrip("n")  # 'n' = the group of reads with an ambiguous base call
    extra = 8 * (int(w) - 1) if w.isdigit() and (base.startswith("k_pass") or base.startswith("k_collapse") or base.startswith("k_cascade")) else 0
    if base.startswith("k_cascade") and not per_pass:
        return CASCADE_BYTES_PER_READ + extra
    return (ALGO_BYTES["k_pass"] if base.startswith(("k_pass", "k_cascade_bulk")) else ALGO_BYTES.get(base, 0)) + extra


def group_index(rec_name):
    """storage group of a profile record ('.w2n' -> 2-word reads with an N): index into DeviceReads.group_counts()"""
    _, _, w = rec_name.partition(".w")
    has_n = w.endswith("n")
    w = w.rstrip("n")
    return ({"1": 0, "2": 1, "4": 2, "8": 3, "16": 4}.get(w, 0), has_n)


def rocprof_symbol(rec_name):
    """profile record 'k_pass[8].w1' -> rocprofv3 kernel name fragment 'k_pass<1, 8>'"""
    base, _, w = rec_name.partition(".w")
    w = w.rstrip("n") or "1"
    if base.startswith("k_pass["):
        return f"k_pass<{w}, {int(base[7:-1].split('-')[0])}>"  # a merged run 'k_pass[4-6]' is launched as slot 4
    if base in ("k_cascade_fused", "k_cascade_bulk"):  # <W, HASN>: the record's 'n' suffix is the group with N masks
        return f"{base}<{w}, {'true' if rec_name.endswith('n') else 'false'},"  # (<W, HASN, REP>: either build of the group's kernel)
    if base in ("k_collapse_insert", "k_collapse_scatter"):
        return f"{base}<{w}>"
    if base == "k_part_dedup":  # k_part_dedup<2048> / <4096>: one of them per sample size
        return "k_part_dedup<"
    if base == "k_join":  # the record covers the rows kernel (both launches) or the atomic forms
        return "k_join_rows("
    return base + "("


def stage_rooflines(kernels, n_steps, raw_groups, uniq_groups, S, n_mirna, dom_roofline, tally=None):
    """SURVEY.md 8(d) 'Algorithmic bytes' / BASELINE.md 3.4 price EVERY stage of the step, not only the dominant kernel:
    collapse 9 B in per raw read + 13 B out per unique (+ 8 B per further word of a wider read); cascade 14 B per collapsed
    read (the figure of `roofline`); count join (5 + 4 S) B in per collapsed read + 8 (2 R_mirna + 10) S B out; C5 tally
    (9 + 4 S) B in per miRNA read + R_mirna * 30 * 12 * 8 B out.  Times: HIP events around every launch in the bracketed
    warm-up steps of the same batch (`kernels`), summed over the BULK read group's kernels of the stage -- the small groups'
    run beside them on streams of their own -- so each fraction is what the stage's critical path achieves."""
    half = len(raw_groups) // 2
    gi = int(np.argmax(raw_groups))
    width = [1, 2, 4, 8, 16][gi % half]
    sfx = f".w{width}" + ("n" if gi >= half else "")
    extra = 8 * (width - 1)

    def ms_of(pred):
        names = [k for k in kernels if pred(k)]
        return sum(kernels[k]["avg_ms"] * kernels[k]["launches"] for k in names) / n_steps, names

    out = {}
    coll_ms, coll_k = ms_of(lambda k: k.endswith(sfx) and k.startswith(("k_part_", "k_collapse_", "k_heads")))
    n_raw, n_u = float(raw_groups[gi]), float(uniq_groups[gi])
    if coll_ms > 0:
        b = (9 + extra) * n_raw + (13 + extra) * n_u
        out["collapse"] = {"kernels": sorted(coll_k), "ms": round(coll_ms, 5), "algorithmic_bytes": round(b, 1),
                           "achieved": round(b / (coll_ms * 1e-3) / 1e9, 2), "frac": round(b / (coll_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                           "units": {"raw_reads": n_raw, "unique_reads": n_u},
                           "bytes_per_unit": f"{9 + extra} in per raw read + {13 + extra} out per unique read"}
    out["cascade"] = {"kernels": [dom_roofline["kernel"]], "ms": dom_roofline["avg_launch_ms"],
                      "algorithmic_bytes": round(dom_roofline["algorithmic_bytes_per_unit"] * dom_roofline["units_per_launch"], 1),
                      "achieved": dom_roofline["achieved"], "frac": dom_roofline["frac"], "bytes_per_unit": "see `roofline`"}
    join_ms, join_k = ms_of(lambda k: k.startswith("k_join"))
    n_all = float(sum(uniq_groups))
    if join_ms > 0:
        b = (5 + 4 * S) * n_all + 8 * (2 * n_mirna + 10) * S
        out["join"] = {"kernels": sorted(join_k), "ms": round(join_ms, 5), "algorithmic_bytes": round(b, 1),
                       "achieved": round(b / (join_ms * 1e-3) / 1e9, 2), "frac": round(b / (join_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                       "units": {"unique_reads": n_all, "samples": S, "mirna_references": n_mirna},
                       "bytes_per_unit": f"{5 + 4 * S} in per collapsed read (all groups) + {8 * (2 * n_mirna + 10) * S} out"}
    tally_ms, tally_k = ms_of(lambda k: k in ("k_member_list", "k_tally"))
    if tally is not None and tally_ms > 0:
        b = (9 + 4 * S) * float(tally) + n_mirna * 30 * 12 * 8
        out["c5_tally"] = {"kernels": sorted(tally_k), "ms": round(tally_ms, 5), "algorithmic_bytes": round(b, 1),
                           "achieved": round(b / (tally_ms * 1e-3) / 1e9, 2), "frac": round(b / (tally_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                           "units": {"mirna_reads": float(tally)},
                           "bytes_per_unit": f"{9 + 4 * S} in per miRNA read + {n_mirna * 30 * 12 * 8} out"}
    for v in out.values():
        v["unit"], v["peak"], v["bound"] = "GB/s", HBM_PEAK_GBS, "hbm"
    return out


def pmc_traffic(args, rec_name):
    """HBM bytes per launch of one kernel from the PMC counters, as MI355X_MICROARCH.md prescribes:
    FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (no trace domains), kB units, and the gfx950
    correction (FETCH_SIZE counts 128-B requests as 64 B -> x2); a third pass counts SQ_INSTS_VALU (wave-level vector
    instructions) for the secondary, VALU-issue bound of SURVEY 8(d).  Child processes; None on any failure."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3")
    if rocprof is None:
        return None
    # never nest: under a profiler its preloaded tool library would be inherited by the children
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None
    frag = rocprof_symbol(rec_name)
    out = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
        d = tempfile.mkdtemp(prefix="mirge_pmc_", dir="/tmp")
        # rocprofv3 is a python script: run it with this interpreter (no '#!/usr/bin/env' hop)
        cmd = [sys.executable, rocprof, "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-baseline", "0", "--pmc", "0",
               "--min-seconds", "0", "--spinup", "0", "--cli-path", "0", "--two-in-flight", "0", "--read-sets", "0",
               "--reads", str(args.reads), "--scale", args.scale, "--workload", args.workload, "--pool", str(args.pool)]
        try:
            subprocess.run(cmd, timeout=900, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if r["Counter_Name"] == ctr and frag in r["Kernel_Name"]:
                            vals.append((int(r["Grid_Size"]), float(r["Counter_Value"])))
            if not vals:
                if ctr == "SQ_INSTS_VALU":  # the traffic stands without the secondary bound
                    continue
                return None
            gmax = max(g for g, _ in vals)  # the big read group's launches (the N / long-read groups share the symbol)
            sel = [v for g, v in vals if g == gmax]
            out[ctr] = sum(sel) / len(sel)
        except Exception:
            if ctr == "SQ_INSTS_VALU":
                continue
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"fetch_kb_raw": out["FETCH_SIZE"], "write_kb": out["WRITE_SIZE"], "valu_insts": out.get("SQ_INSTS_VALU"),
            "bytes": (2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0,
            "bytes_raw": (out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0}


def reference_bowtie_dir():
    """directory of a real bowtie 1.x (bowtie + bowtie-build), or None"""
    import shutil
    d = os.environ.get("MIRGE_BOWTIE_DIR")
    if d and os.path.exists(os.path.join(d, "bowtie-build")):
        return d
    b = shutil.which("bowtie-build")
    if b and shutil.which("bowtie") and os.path.dirname(os.path.realpath(b)) != os.path.join(ROOT, "mirge3.0_amd", "shim"):
        return os.path.dirname(b)
    return None


def reference_bowtie_baseline(bowtie_dir, sl, libs, reads, n_pass, ctx, casc, args):
    """`cpu_baseline.kind = "reference"`: the reference's own ten bowtie runs (manifoldAlign.py:85,92-135) on the
    collapsed reads of a bounded sample, all host cores, index construction (bowtie-build) not timed; then the
    per-pass membership of bowtie vs the GPU engine on the same reads (tools/bowtie_crosscheck.py's comparison)."""
    import re
    import shlex
    import subprocess
    import tempfile
    from collections import Counter
    from mirge3_amd import _ffi
    from mirge3_amd.cascade import PASSES
    from mirge3_amd.seqio import FlatSeqs, index_basename, write_fasta
    cores = os.cpu_count() or 1
    tmp = tempfile.mkdtemp(prefix="mirge_refbowtie_", dir="/tmp")
    idx = {}
    for key, lib in libs.items():
        base = os.path.join(tmp, index_basename("bench", key, "miRBase"))
        write_fasta(base + ".fa", lib)
        subprocess.run([os.path.join(bowtie_dir, "bowtie-build"), "-q", "--threads", str(cores), base + ".fa", base],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        idx[key] = base
    m = min(len(reads), max(1000, min(args.cpu_sample, 2_000_000)))
    cnt = Counter(reads.take(np.arange(m)).to_list())
    seqs = list(cnt)
    ver = subprocess.run([os.path.join(bowtie_dir, "bowtie"), "--version"], capture_output=True, text=True).stdout.split("\n")[0]
    fasta = os.path.join(tmp, "bwtInput.fasta")
    annotated, per_pass, t_total, differ = {}, [], 0.0, 0
    for it in range(n_pass):
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
        t = time.perf_counter()
        o = subprocess.run([os.path.join(bowtie_dir, "bowtie"), idx[key]] + shlex.split(argstr) + [str(cores), fasta],
                           check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True).stdout
        dt = time.perf_counter() - t
        t_total += dt
        hit = {}
        for ln in o.split("\n"):
            if ln and not ln.startswith("@"):
                f = ln.split("\t")
                if f[2] != "*":
                    hit[f[0]] = f[2]
        got = set()
        if recs:
            dr = _ffi.DeviceReads.pack(ctx, FlatSeqs.from_list([q for q, _ in recs]))
            res = _ffi.cascade_run(ctx, dr, [casc.dev_libs[it]], [casc.policies[it]])
            ps = res.fetch()[0]
            got = {q for (q, _), p in zip(recs, ps) if p == 0}
            res.close(); dr.close()
        differ += len(set(hit) ^ got)
        per_pass.append({"pass": it, "reads": len(recs), "bowtie_aligned": len(hit), "gpu_aligned": len(got),
                         "membership_differs": len(set(hit) ^ got), "s": round(dt, 3)})
        for q in hit:
            annotated[q] = it
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return {"value": round(m / max(t_total, 1e-9) / 1e6, 4), "unit": "M reads/s", "cores": cores, "kind": "reference",
            "sample": f"{ver}: the reference's bowtie argument strings verbatim, --threads {cores}, on the {len(seqs)} collapsed reads "
                      f"of the first {m} raw reads of the same sample; {t_total:.2f} s of bowtie (bowtie-build and the Python "
                      f"around it not timed)",
            "per_pass": per_pass,
            "parity": ("alignment predicate PINNED on this box: per-pass aligned/unaligned membership of bowtie vs the GPU engine "
                       f"differs for {differ} reads" if differ else
                       "alignment predicate PINNED on this box: per-pass membership identical to " + ver)}


def gpu_clocks_mhz():
    """current shader clock of every GPU the driver exposes under /sys (the line pp_dpm_sclk marks with '*'), or None"""
    import glob
    import re
    out = {}
    for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))[:16]:
        try:
            with open(f) as fh:
                for line in fh:
                    if line.rstrip().endswith("*"):
                        m = re.search(r"(\d+)\s*[Mm][Hh][Zz]", line)
                        if m:
                            out[f.split("/")[4]] = int(m.group(1))
        except OSError:
            pass
    return out or None


def fastq_text(reads):
    """4-line FASTQ records (quality 'I') of a FlatSeqs as one uint8 array, built with numpy"""
    L = reads.lengths
    rec = 2 * L + 6  # "@\n" seq "\n+\n" qual "\n"
    roff = np.zeros(len(reads) + 1, dtype=np.int64)
    np.cumsum(rec, out=roff[1:])
    text = np.full(int(roff[-1]), ord("I"), dtype=np.uint8)
    text[roff[:-1]] = ord("@")
    text[roff[:-1] + 1] = 10
    rows = np.repeat(np.arange(len(reads), dtype=np.int64), L)
    within = np.arange(int(reads.offsets[-1]), dtype=np.int64) - reads.offsets[:-1][rows]
    text[roff[:-1][rows] + 2 + within] = reads.data
    del rows, within
    text[roff[:-1] + 2 + L] = 10
    text[roff[:-1] + 3 + L] = ord("+")
    text[roff[:-1] + 4 + L] = 10
    text[roff[1:] - 1] = 10
    return text


def write_fastq_file(reads, path, chunk=2_000_000):
    """the sample as a FASTQ file, built 2 M reads at a time (fastq_text takes ~0.5 KB of host memory per read: eight ranks writing
    20 M-read samples at once would want 80 GB)"""
    from mirge3_amd.seqio import FlatSeqs
    off = reads.offsets
    with open(path, "wb") as fh:
        for a in range(0, len(reads), chunk):
            b = min(len(reads), a + chunk)
            fastq_text(FlatSeqs(reads.data[int(off[a]):int(off[b])], off[a:b + 1] - off[a])).tofile(fh)
    return os.path.getsize(path)


class _GroupDist:
    """torch.distributed's object collectives bound to ONE process group (a gloo group with a short timeout of its own: a rank
    that fails inside the leg must cost the bench line two minutes, not the default group's ten)"""
    def __init__(self, dist, group):
        self._d, self._g = dist, group

    def all_gather_object(self, out, obj):
        return self._d.all_gather_object(out, obj, group=self._g)

    def gather_object(self, obj, out=None, dst=0):
        return self._d.gather_object(obj, out, dst=dst, group=self._g)

    def broadcast_object_list(self, box, src=0):
        return self._d.broadcast_object_list(box, src=src, group=self._g)

    def barrier(self):
        return self._d.barrier(group=self._g)


def c4_end_to_end(args, dist, rank, world, sl, casc, reads, n_pass):
    """BASELINE configs[3] as the PRODUCT runs it (never `value`): N FASTQ files on disk, one per rank -> every output file of the
    sharded CLI, the run's ONE mapped.csv / unmapped.csv over the sorted union of all samples included (digest.py:243,
    mirge/__main__.py:164-173).  `value` steps collapse -> cascade -> join per rank and never merges; this leg is the wall time
    between two barriers around what `python -m torch.distributed.run ... -m mirge3_amd.cli -s S0.fastq,...` does with the libraries
    resident: parse, collapse + cascade + join per rank, then the tail -- round 6: every rank merges, annotates, orders, formats and
    pwrites ITS key range of the two files (fastpath.run_sharded_ranges); for comparison the same samples with rank 0 building the
    joint table alone (round 5's route, MIRGE_SHARD_TAIL=rank0)."""
    import datetime
    import shutil
    import tempfile
    from types import SimpleNamespace
    from mirge3_amd import fastpath, multigpu
    if n_pass != 9:
        return None
    pg = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=180))
    gd = _GroupDist(dist, pg)
    box = [tempfile.mkdtemp(prefix="mirge_c4e2e_", dir="/tmp") if rank == 0 else None]
    gd.broadcast_object_list(box, src=0)
    tmp = box[0]
    out = {}
    try:
        if rank == 0:
            os.makedirs(os.path.join(tmp, "Libs", "bench", "annotation.Libs"))
            with open(os.path.join(tmp, "Libs", "bench", "annotation.Libs", "bench_merges_miRBase.csv"), "w") as fh:
                fh.write("".join(",".join(r) + "\n" for r in sl.merges))
        fq = os.path.join(tmp, f"S{rank}.fastq")
        fq_bytes = write_fastq_file(reads, fq)
        names = [f"S{i}" for i in range(world)]
        files = [os.path.join(tmp, f"S{i}.fastq") for i in range(world)]
        a = SimpleNamespace(libraries_path=os.path.join(tmp, "Libs"), organism_name="bench", spikeIn=False, quiet=True, minimum_length=16,
                            crThreshold="0.1", device=casc.ctx.device, isoform_entropy=False)

        def one_run(label, ranges):
            work = os.path.join(tmp, label)
            if rank == 0:
                os.makedirs(work)
            gd.barrier()
            t0 = time.perf_counter()
            held = {} if ranges else None
            tb = fastpath.run_sample_tables(a, files[rank], names[rank], rank, work, "miRBase", casc, True, False, hold=held)
            t_sample = time.perf_counter() - t0
            tables = multigpu.gather_tables([tb], rank, world, gd)
            tm = {}
            if ranges:
                tm = fastpath.run_sharded_ranges(a, held, world, names, work, casc, rank, world, gd)
                if rank == 0:
                    fastpath.sharded_count_tables(a, tables, work, "miRBase", casc, tm)
            elif rank == 0:
                o = fastpath.run_sharded_rank0(a, tables, work, "miRBase", casc, timings=tm)
                for h in ("uniq", "res"):
                    if o["device"].get(h) is not None:
                        o["device"][h].close()
            gd.barrier()
            wall = time.perf_counter() - t0
            every = [None] * world
            gd.all_gather_object(every, (wall, t_sample, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in tm.items()
                                                         if not isinstance(v, (list, dict))}))
            res = {"wall_s": round(max(x[0] for x in every), 4), "M_reads_per_s": round(world * args.reads / max(x[0] for x in every) / 1e6, 2),
                   "sample_s_per_rank": [round(x[1], 4) for x in every],
                   "tail_s": round(max(x[0] for x in every) - max(x[1] for x in every), 4)}
            if ranges:
                res["range_tail_per_rank"] = [x[2] for x in every]
            else:
                res["rank0_tail"] = every[0][2]
            if rank == 0:
                res["output_bytes"] = {f: os.path.getsize(os.path.join(work, f)) for f in ("mapped.csv", "unmapped.csv", "miR.Counts.csv")}
            return res, work

        def drop(work):  # (a run's two per-read files are GBs at C4's size: only the two runs that are compared stay side by side)
            gd.barrier()
            if rank == 0:
                shutil.rmtree(work, ignore_errors=True)
        first, w_f = one_run("ranges_first", True)   # pays for page-locked staging, buffer-pool blocks, the page cache
        drop(w_f)
        out["ranges"], w_r = one_run("ranges", True)
        out["ranges"]["first_run_wall_s"] = first["wall_s"]
        _, w_f0 = one_run("rank0_first", False)
        drop(w_f0)
        out["rank0_alone"], w_0 = one_run("rank0", False)
        if rank == 0:
            import subprocess
            out["same_files_both_tails"] = all(subprocess.run(["cmp", "-s", os.path.join(w_r, f), os.path.join(w_0, f)]).returncode == 0
                                               for f in ("mapped.csv", "unmapped.csv", "miR.Counts.csv", "miR.RPM.csv", "annotation.report.csv"))
        out["fastq_MB_per_sample"] = round(fq_bytes / 1e6, 1)
        out["speedup_over_rank0_alone"] = round(out["rank0_alone"]["wall_s"] / max(out["ranges"]["wall_s"], 1e-9), 3)
        out["note"] = ("wall time between two barriers: every rank's FASTQ file on disk -> parse + collapse + cascade + join on its GPU -> the run's "
                       "count tables and ONE mapped.csv / unmapped.csv (sorted union of all samples, one count column each); libraries resident; "
                       "`ranges` = every rank writes its key range of the two files, `rank0_alone` = round 5's route; second run of each (the first "
                       "pays for page-locked staging and pool blocks); tail_s = wall - the slowest rank's own sample; never `value`")
    finally:
        try:
            gd.barrier()
        except Exception:  # noqa: BLE001
            pass
        if rank == 0:
            shutil.rmtree(tmp, ignore_errors=True)
    return out


def write_gzip_level6(text, path, piece=4 << 20):
    """`text` as ONE gzip member at level 6, compressed in independent pieces on all cores (pigz -i's construction: every
    piece its own raw deflate stream ended with a sync flush, the last one with the final block): what a user's
    sample.fastq.gz is to a reader -- a single inflate stream -- without 30 s of single-threaded compression in the bench."""
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    mv = memoryview(text)
    n = len(mv)
    cuts = list(range(0, n, piece)) or [0]

    def deflate(i):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        last = i == cuts[-1]
        return co.compress(mv[i:i + piece]) + co.flush(zlib.Z_FINISH if last else zlib.Z_SYNC_FLUSH)

    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as pool:
        parts = list(pool.map(deflate, cuts))
    crc = 0
    for i in cuts:
        crc = zlib.crc32(mv[i:i + piece], crc)
    with open(path, "wb") as fh:
        fh.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
        for p in parts:
            fh.write(p)
        fh.write((crc & 0xFFFFFFFF).to_bytes(4, "little") + (n & 0xFFFFFFFF).to_bytes(4, "little"))
    return os.path.getsize(path)


def secondary_read_sets(args, sl, ctx, casc, _ffi, synth, n_mirna, EXACT_PASS, ISO_PASS, out):
    """Short legs on the two read sets SURVEY 8(d) specifies: same libraries, same step (mirge_collapse_cascade + count join), ~0.3 s
    of steps each after 3 warm-up steps."""
    res = {"default_draw": {"raw_reads": args.reads, "unique_reads": out["config"]["unique_reads_per_gpu"],
                            "U_over_N": round(out["config"]["unique_reads_per_gpu"] / args.reads, 4), "ms_per_step": out["ms_per_step"],
                            "M_raw_reads_per_s": out["value"], "M_collapsed_reads_per_s": out["collapsed_reads_per_s_M"],
                            "note": "`value`: every read drawn independently from the class mix"}}
    n = args.reads

    from mirge3_amd.cascade import Cascade as _Cascade

    def in_flight(reads_fs, one_ms, counts=(2, 4)):
        """the same sample stepped by k contexts side by side (k host threads, k sets of HIP streams, k copies of the libraries): what
        a batch of such samples per GPU runs at -- on the default draw the chip is full and two contexts only contend (0.83-0.86 x);
        this set leaves it mostly idle"""
        import threading
        res, extra = {"1": {"ms_per_sample": round(one_ms, 4)}}, []
        try:
            for _ in range(max(counts) - 1):
                c2 = _ffi.Context(casc.ctx.device)
                k2 = _Cascade(c2, casc.libs, n_pass=9)
                extra.append((c2, k2, _ffi.DeviceReads.pack(c2, reads_fs)))
            r0 = _ffi.DeviceReads.pack(ctx, reads_fs)
            workers = [(ctx, casc, r0)] + extra

            def stepper(w, reps):
                cx, kc, rw = w
                for _ in range(reps):
                    uq, rs = kc.collapse_and_run(rw)
                    _ffi.count_join(cx, uq, rs, EXACT_PASS, ISO_PASS, n_mirna)
                    rs.close(); uq.close()
            for w in workers:
                stepper(w, 3)
            reps = max(20, min(400, int(0.3 / max(one_ms * 1e-3, 1e-6))))
            for kk in counts:
                th = [threading.Thread(target=stepper, args=(w, reps)) for w in workers[:kk]]
                t = time.perf_counter()
                for x in th:
                    x.start()
                for x in th:
                    x.join()
                per = (time.perf_counter() - t) / (kk * reps)
                res[str(kk)] = {"ms_per_sample": round(per * 1e3, 4), "speedup": round(one_ms / (per * 1e3), 3)}
            r0.close()
        finally:
            for c2, k2, rw in extra:
                rw.close(); k2.close(); c2.close()
        return res

    def leg(reads_fs, note, flights=None):
        raw_s = _ffi.DeviceReads.pack(ctx, reads_fs)
        u_n = [0]

        def one():
            uq, rs = casc.collapse_and_run(raw_s)
            _ffi.count_join(ctx, uq, rs, EXACT_PASS, ISO_PASS, n_mirna)
            u_n[0] = len(uq)
            rs.close(); uq.close()
        for _ in range(3):
            one()
        k = 0
        t0 = time.perf_counter()
        while k < 20 or time.perf_counter() - t0 < 0.3:
            one(); k += 1
        dt = (time.perf_counter() - t0) / k
        raw_s.close()
        out_l = {"raw_reads": len(reads_fs), "unique_reads": u_n[0], "U_over_N": round(u_n[0] / len(reads_fs), 4), "steps": k,
                 "ms_per_step": round(dt * 1e3, 4), "M_raw_reads_per_s": round(len(reads_fs) / dt / 1e6, 1),
                 "M_collapsed_reads_per_s": round(u_n[0] / dt / 1e6, 1), "note": note}
        if flights:
            try:
                out_l["samples_in_flight"] = in_flight(reads_fs, dt * 1e3, flights)
            except Exception as e:  # noqa: BLE001
                out_l["samples_in_flight"] = {"error": repr(e)[:300]}
        return out_l

    pool = max(1000, n // 8)

    def multi_sample(S=4):
        """SEVERAL samples in one process and ONE launch set (never `value`): what the CLI runs for `-s S1.fastq,S2.fastq,...` -- every sample
        collapsed by itself, the dictionaries merged on the device (mirge_collapse_merge, round 6), one cascade and one count join over the
        union with S count columns -- against round 5's route (mirge_collapse with sample ids over the samples' raw reads: the general,
        global-atomic path) and against S separate steps.  Four samples of the Zipf kind."""
        smp = [synth.make_reads(sl, n, seed=2000 + s, pool=pool) for s in range(S)]
        raws = [_ffi.DeviceReads.pack(ctx, x) for x in smp]
        del smp
        sid = np.repeat(np.arange(S, dtype=np.int32), [len(r) for r in raws])
        nu = [0]

        def merged():
            dicts = [r.collapse() for r in raws]
            u = _ffi.DeviceReads.merge(ctx, dicts)
            for d in dicts:
                d.close()
            rs = casc.run(u)
            _ffi.count_join(ctx, u, rs, EXACT_PASS, ISO_PASS, n_mirna)
            nu[0] = len(u)
            rs.close(); u.close()

        def joint_raw():
            allr = _ffi.DeviceReads.concat(ctx, raws)
            u = allr.collapse(sid, S)
            rs = casc.run(u)
            _ffi.count_join(ctx, u, rs, EXACT_PASS, ISO_PASS, n_mirna)
            rs.close(); u.close(); allr.close()

        def separate():
            for r in raws:
                u, rs = casc.collapse_and_run(r)
                _ffi.count_join(ctx, u, rs, EXACT_PASS, ISO_PASS, n_mirna)
                rs.close(); u.close()
        out_m = {}
        for name, f in (("merged_dictionaries", merged), ("joint_collapse_of_raw_reads", joint_raw), ("separate_steps", separate)):
            f(); f()
            k, t0 = 0, time.perf_counter()
            while k < 5 or time.perf_counter() - t0 < 0.3:
                f(); k += 1
            dt = (time.perf_counter() - t0) / k
            out_m[name] = {"ms_per_run": round(dt * 1e3, 4), "ms_per_sample": round(dt * 1e3 / S, 4), "M_raw_reads_per_s": round(S * n / dt / 1e6, 1)}
        for r in raws:
            r.close()
        out_m.update(samples=S, raw_reads_per_sample=n, unique_reads_of_the_union=nu[0],
                     note="S samples, one process: merged_dictionaries = the CLI's route since round 6 (per-sample partitioned collapses + "
                          "mirge_collapse_merge + one cascade + one S-column join); joint_collapse_of_raw_reads = its route until round 5; "
                          "separate_steps = S times the step of `value` (S dictionaries, S cascades, no union)")
        return out_m

    res["zipf_pool"] = leg(synth.make_reads(sl, n, seed=2000, pool=pool),
                           f"Zipf(s = 1.1) duplication over {pool} templates of the same class mix: SURVEY 8(d)'s 'realistic' set", flights=(2, 4))
    # all distinct: the default class mix without its duplicated exact-miRNA class, collapsed once on the GPU, its unique reads
    # taken as the raw reads of the leg (every read once)
    mix = dict(synth.DEFAULT_MIX, exact=0.01)
    cand = synth.make_reads_chunked(sl, int(n * 1.25), seed=3000, mix=mix)
    r0 = _ffi.DeviceReads.pack(ctx, cand)
    u0 = r0.collapse()
    distinct = u0.unpack()
    u0.close(); r0.close()
    del cand
    if len(distinct) > n:
        distinct = distinct.take(np.arange(n))
    try:
        res["multi_sample_run"] = multi_sample(4)
    except Exception as e:  # noqa: BLE001
        res["multi_sample_run"] = {"error": repr(e)[:300]}
    res["distinct"] = leg(distinct, "every read exactly once (U = N): SURVEY 8(d)'s 'stress' set -- the class mix minus its duplicated exact-miRNA "
                                    "reads, de-duplicated; every read goes through the cascade")
    return res


def repeat_rich_leg(args, ctx, _ffi, synth, Cascade, PASSES, n_mirna, EXACT_PASS, ISO_PASS, default_ms):
    """The step on libraries with the repeat structure real ones have (never `value`; round 5's review: every number so far is on
    uniform-random libraries, the best case for a k-mer index): `synth.make_libraries(repeats=True)` -- poly-A tails on 60 % of the
    mRNAs, five 300-nt Alu-like families at 5-15 % divergence in 10 % of the transcripts, simple repeats in ncRNA and mRNA, tRNA
    isodecoder families -- and `synth.REPEAT_MIX` reads (15 % poly-A / poly-T / simple-repeat / Alu-derived reads with 0-2 errors).
    Reports the step, the bulk cascade kernel's own time, how unevenly its workgroups finished (one read whose probe bucket holds 10^5
    positions is scanned by ONE wave), and the oracle's verdict on a 40 k-read sample of the same set."""
    import oracle
    t0 = time.perf_counter()
    sl_r = synth.make_libraries(seed=20260101, scale=args.scale, repeats=True)
    casc_r = Cascade(ctx, sl_r.libs, n_pass=9)
    n = args.reads
    reads_r = synth.make_reads_chunked(sl_r, n, seed=4000, mix=synth.REPEAT_MIX)
    raw_r = _ffi.DeviceReads.pack(ctx, reads_r)
    setup_s = time.perf_counter() - t0
    u_n = [0]

    def one():
        uq, rs = casc_r.collapse_and_run(raw_r)
        _ffi.count_join(ctx, uq, rs, EXACT_PASS, ISO_PASS, n_mirna)
        u_n[0] = len(uq)
        rs.close(); uq.close()
    t1 = time.perf_counter()
    one()
    first_step_s = time.perf_counter() - t1  # probe tables of these libraries are built here
    for _ in range(2):
        one()
    ctx.profile(True); ctx.profile_only(""); ctx.profile_reset()
    for _ in range(3):
        one()
    recs_all = [r for r in ctx.profile_records() if r[1]]
    recs = [r for r in recs_all if r[0].startswith("k_cascade_bulk")]
    wg = _ffi.cascade_wg_times(ctx)
    ctx.profile(False); ctx.profile_only("")
    bulk_ms = max((ms / l for _, l, ms, _ in recs), default=None)
    top = sorted(((nm, ms / l) for nm, l, ms, _ in recs_all), key=lambda x: -x[1])[:6]
    k = 0
    t = time.perf_counter()
    while k < 10 or time.perf_counter() - t < 0.3:
        one(); k += 1
    dt = (time.perf_counter() - t) / k
    out = {"raw_reads": n, "unique_reads": u_n[0], "U_over_N": round(u_n[0] / n, 4), "steps": k, "ms_per_step": round(dt * 1e3, 4),
           "M_raw_reads_per_s": round(n / dt / 1e6, 1), "M_collapsed_reads_per_s": round(u_n[0] / dt / 1e6, 1),
           "vs_default_draw_step": round(dt * 1e3 / default_ms, 3), "k_cascade_bulk_ms": None if bulk_ms is None else round(bulk_ms, 4),
           "setup_s": round(setup_s, 1), "first_step_s": round(first_step_s, 3), "longest_kernels_ms": {nm: round(v, 4) for nm, v in top},
           "library_bases": {kk: v.total_len for kk, v in sl_r.libs.items()},
           "note": "libraries with poly-A tails, Alu-like families, simple repeats and tRNA isodecoder families + 15 % repeat-derived reads; "
                   "vs_default_draw_step = this step / the uniform libraries' step of `value`"}
    if wg is not None and len(wg):
        out["bulk_workgroups_ms"] = {"n": int(len(wg)), "median": round(float(np.median(wg)), 4), "p99": round(float(np.percentile(wg, 99)), 4),
                                     "max": round(float(wg.max()), 4),
                                     "longest_share_of_launch": None if not bulk_ms else round(float(wg.max()) / bulk_ms, 3),
                                     "note": "constant-rate clock, first to last instruction of every workgroup of the last bracketed launch: max / median "
                                             "near 1 = the fixed segments finished together; a long tail = one segment's reads held the launch"}
    # the oracle on a 40 k-read sample of the same set (brute-force semantics, k-mer variant), per read and per class
    try:
        m = min(40_000, n)
        sub = reads_r.take(np.arange(m))
        r2 = _ffi.DeviceReads.pack(ctx, sub)
        u2 = r2.collapse()
        res2 = casc_r.run(u2)
        g = res2.fetch()
        useq = u2.unpack()
        libs_o = [(sl_r.libs[PASSES[p][1]].seqs.data, sl_r.libs[PASSES[p][1]].seqs.offsets) for p in range(9)]
        t = time.perf_counter()
        o = oracle.cascade(useq.data, useq.offsets, libs_o, n_pass=9, indexed=True, threads=min(os.cpu_count() or 1, 64))
        out["oracle_parity_on_sample"] = {"raw_reads": m, "unique_reads": len(u2), "oracle_s": round(time.perf_counter() - t, 2),
                                          "identical": bool(all(np.array_equal(a.astype(np.int64), b.astype(np.int64)) for a, b in zip(o, g))),
                                          "annotated": int((g[0] >= 0).sum())}
        res2.close(); u2.close(); r2.close()
    except Exception as e:  # noqa: BLE001
        out["oracle_parity_on_sample"] = {"error": repr(e)[:300]}
    raw_r.close(); casc_r.close()
    return out


def cli_path(args, sl, libs, text, n_pass):
    """FASTQ file -> all CSVs through the CLI's device-resident route, wall-clock: a first run (libraries read from
    their directory, packed, indexed: what a one-sample invocation pays) and a second one in the same process
    (libraries resident: what every further sample of a batch pays)."""
    import shutil
    import tempfile
    from types import SimpleNamespace
    from mirge3_amd import fastpath
    from mirge3_amd.seqio import index_basename, write_fasta
    if n_pass != 9:
        return None
    tmp = tempfile.mkdtemp(prefix="mirge_cli_", dir="/tmp")
    try:
        idx = os.path.join(tmp, "Libs", "bench", "index.Libs")
        os.makedirs(idx)
        os.makedirs(os.path.join(tmp, "Libs", "bench", "annotation.Libs"))
        for key, lib in libs.items():
            write_fasta(os.path.join(idx, index_basename("bench", key, "miRBase") + ".fa"), lib)
        with open(os.path.join(tmp, "Libs", "bench", "annotation.Libs", "bench_merges_miRBase.csv"), "w") as fh:
            fh.write("".join(",".join(r) + "\n" for r in sl.merges))
        fq = os.path.join(tmp, "S1.fastq")
        text.tofile(fq)
        a = SimpleNamespace(libraries_path=os.path.join(tmp, "Libs"), organism_name="bench", spikeIn=False, quiet=True,
                            minimum_length=16, crThreshold="0.1", device=0, isoform_entropy=False)
        res = {}
        from mirge3_amd import cascade as _casc
        old_cache = os.environ.get("MIRGE_LIB_CACHE")
        os.environ["MIRGE_LIB_CACHE"] = "1"  # the first run writes <index>.mirge3amd next to the FASTA files it read
        for label in ("first_run", "libraries_resident", "new_process_cached_libraries"):
            work = os.path.join(tmp, label)
            os.makedirs(work)
            if label == "new_process_cached_libraries":  # what a later one-sample invocation pays: the libraries' packed
                for cc in _casc._cascade_cache.values():  # images come from the cache, their tables are rebuilt on the device
                    cc.close()
                _casc._cascade_cache.clear()
            tm = {}
            t = time.perf_counter()
            o = fastpath.run(a, [fq], ["S1"], work, "miRBase", timings=tm)
            wall = time.perf_counter() - t
            for h in ("uniq", "res"):
                o["device"][h].close()
            sizes = {f: os.path.getsize(os.path.join(work, f)) for f in ("mapped.csv", "unmapped.csv", "miR.Counts.csv")}
            res[label] = {"wall_s": round(wall, 3), "M_reads_per_s": round(args.reads / wall / 1e6, 2),
                          "stages_s": {k: (round(v, 3) if not isinstance(v, dict) else v) for k, v in tm.items()}, "output_bytes": sizes}
        # ---- the same sample with -gff: fetch of reads + annotation, k_isotype, sample_miRge3.gff on the host's cores
        try:
            mir, hp = sl.libs["mirna"], sl.libs["hairpin"]
            os.makedirs(os.path.join(tmp, "Libs", "bench", "fasta.Libs"), exist_ok=True)
            with open(os.path.join(tmp, "Libs", "bench", "fasta.Libs", "bench_mature_miRBase.fa"), "w") as fh:
                fh.write("".join(f">{nm}\n{sq}\n" for nm, sq in zip(mir.names, mir.seqs.to_list())))
            by_hp = {}
            for k, nm in enumerate(mir.names):
                by_hp.setdefault(int(sl.mir_hairpin[k]), []).append(nm)
            with open(os.path.join(tmp, "Libs", "bench", "annotation.Libs", "bench_miRBase.gff3"), "w") as fh:
                for h, names in by_hp.items():
                    fh.write(f"chr1\t.\tmiRNA_primary_transcript\t1\t100\t.\t+\t.\tID=MI{h};Alias=MI{h};Name={hp.names[h]}\n")
                    for nm in names:
                        fh.write(f"chr1\t.\tmiRNA\t1\t22\t.\t+\t.\tID=MIMAT;Alias=MIMAT;Name={nm};Derives_from=MI{h}\n")
            work = os.path.join(tmp, "gff")
            os.makedirs(work)
            a.gff_out = True
            tm = {}
            t = time.perf_counter()
            o = fastpath.run(a, [fq], ["S1"], work, "miRBase", timings=tm)
            wall = time.perf_counter() - t
            for h in ("uniq", "res"):
                o["device"][h].close()
            res["gff_libraries_resident"] = {
                "wall_s": round(wall, 3), "stages_s": {k: (round(v, 3) if not isinstance(v, (dict, list)) else v) for k, v in tm.items()},
                "gff_MB": round(os.path.getsize(os.path.join(work, "sample_miRge3.gff")) / 1e6, 1),
                "note": "the sample of libraries_resident again with -gff: sample_miRge3.gff chosen, typed (k_isotype) and formatted (k_gff_line) on the "
                        "device, its text written with positional writes (mirge_gff_write_device, round 6; round 5: 0.275 s with the records "
                        "and every read, count and annotation fetched and the file built on host cores)"}
        except Exception as e:  # noqa: BLE001
            res["gff_libraries_resident"] = {"error": repr(e)[:300]}
        finally:
            a.gff_out = False
        # ---- the input users hold: the same sample as sample.fastq.gz (one gzip member, level 6), libraries resident
        fq_gz = os.path.join(tmp, "S1.fastq.gz")
        gz_bytes = write_gzip_level6(text, fq_gz)
        import gzip as _gzip
        t = time.perf_counter()
        with _gzip.open(fq_gz, "rb") as fh:
            n_whole = len(fh.read())
        whole_s = time.perf_counter() - t
        first_gz_wall = None
        for sub in ("gz_first", "gz"):  # the second .gz sample of the process finds the text buffer of the first (kept, its pages touched)
            work = os.path.join(tmp, sub)
            os.makedirs(work)
            tm = {}
            t = time.perf_counter()
            o = fastpath.run(a, [fq_gz], ["S1"], work, "miRBase", timings=tm)
            wall = time.perf_counter() - t
            for h in ("uniq", "res"):
                o["device"][h].close()
            if first_gz_wall is None:
                first_gz_wall = wall
        same = all(open(os.path.join(work, f), "rb").read() == open(os.path.join(tmp, "libraries_resident", f), "rb").read()
                   for f in ("mapped.csv", "unmapped.csv", "miR.Counts.csv", "annotation.report.csv"))
        legs = {}
        for mode, sub in (("0", "gz_stream"), ("whole", "gz_whole")):
            os.environ["MIRGE_GZ_PARALLEL"] = mode
            try:
                work2 = os.path.join(tmp, sub)
                os.makedirs(work2)
                tm2 = {}
                t = time.perf_counter()
                o2 = fastpath.run(a, [fq_gz], ["S1"], work2, "miRBase", timings=tm2)
                legs[mode] = (time.perf_counter() - t, tm2)
                for h in ("uniq", "res"):
                    o2["device"][h].close()
            finally:
                os.environ.pop("MIRGE_GZ_PARALLEL", None)
        wall2, tm2 = legs["0"]
        wall3, tm3 = legs["whole"]
        res["gz_libraries_resident"] = {
            "wall_s": round(wall, 3), "M_reads_per_s": round(args.reads / wall / 1e6, 2), "gz_MB": round(gz_bytes / 1e6, 1),
            "text_MB": round(n_whole / 1e6, 1), "same_files_as_plain_fastq": bool(same), "first_gz_sample_of_the_process_wall_s": round(first_gz_wall, 3),
            "stages_s": {k: (round(v, 3) if not isinstance(v, (dict, list)) else v) for k, v in tm.items()},
            "python_gzip_read_whole_s": round(whole_s, 3),
            "streamed_zlib": {"wall_s": round(wall2, 3), "stages_s": {k: (round(v, 3) if not isinstance(v, (dict, list)) else v) for k, v in tm2.items()}},
            "inflated_whole_then_parsed": {"wall_s": round(wall3, 3), "stages_s": {k: (round(v, 3) if not isinstance(v, (dict, list)) else v) for k, v in tm3.items()}},
            "note": "sample.fastq.gz -> every output file.  stages_s.gz_parallel: the member inflated on all host cores by mirge_gz_inflate_progress "
                    "(cut at deflate block starts found by search, decoded without history, resolved, verified against the file's CRC-32) "
                    "on a thread of its own WHILE the main thread uploads and parses the whole records of the text that is already final "
                    "(collapse.ParallelGzipStream; stages_s.gz_stream: inflate_s = the inflation's wall time, inflate_wait_s = the GPU side "
                    "waiting for text, upload_parse_s = its work); wall_s is the process's second .gz sample -- the text buffer of the first is kept, "
                    "its pages touched -- and first_gz_sample_of_the_process_wall_s the first.  inflated_whole_then_parsed: MIRGE_GZ_PARALLEL=whole -- the same "
                    "inflater, the parse behind it (the first form of round 4; the inflation is part of read_files_s).  "
                    "streamed_zlib: the same file with MIRGE_GZ_PARALLEL=0 -- "
                    "the file is inflated by zlib on a worker thread in 8 MB record-aligned "
                    "pieces while the main thread uploads and parses the piece before (collapse.GzipRecordStream); stages_s.gz_stream: "
                    "inflate_s = the worker's time in zlib (the critical path), upload_parse_s = the GPU side's work, inflate_wait_s = "
                    "the GPU side waiting for text; python_gzip_read_whole_s = gzip.open().read() of the same file, what round 3 did "
                    "in front of the parse"}
        res["note"] = ("mirge3_amd.fastpath.run = what `python -m mirge3_amd.cli` executes: FASTQ file read from disk, parsed / "
                       "collapsed / annotated / joined on the GPU, per-miRNA tables by pandas on ~2.7 k rows, mapped.csv + "
                       "unmapped.csv (one line per unique read) formatted on the GPU (mirge_annotation_csv_device); first_run reads "
                       "the libraries' FASTA and writes their cache, new_process_cached_libraries drops the resident libraries and "
                       "loads the cache (what the next invocation pays); not part of `value`")
        return res
    finally:
        if "old_cache" in locals():
            os.environ.pop("MIRGE_LIB_CACHE", None) if old_cache is None else os.environ.__setitem__("MIRGE_LIB_CACHE", old_cache)
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=["c3", "c2", "c4", "c5"],
                    help="default: c3 (BASELINE configs[2]) at N = 1, c4 (configs[3]: one 20 M-read sample per GPU) under torch.distributed.run")
    ap.add_argument("--reads", type=int, default=None, help="raw reads per sample (default 10 M; 20 M for c4)")
    ap.add_argument("--scale", default="full", choices=["ci", "small", "full"])
    ap.add_argument("--two-calls", type=int, default=0, help="1: mirge_collapse then mirge_cascade_run instead of the one-call path")
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--cpu-sample", type=int, default=10_000_000)
    ap.add_argument("--long-frac", dest="long_frac", type=float, default=None,
                    help="share of the sample drawn as 32-50-nt reads (synth default 0.05); a side measurement, labelled in config.workload")
    ap.add_argument("--pool", type=int, default=0,
                    help="draw the reads from this many templates with Zipf weights (SURVEY 8d 'realistic' U/N); 0 = independent draws")
    ap.add_argument("--cli-path", type=int, default=1, help="rank 0, N=1: also time the CLI's route from a FASTQ file on disk to all CSVs")
    ap.add_argument("--min-seconds", dest="min_seconds", type=float, default=2.0,
                    help="repeat the timed region of --steps steps until this much timed work has been seen (0 = one region)")
    ap.add_argument("--spinup", type=float, default=0.3, help="seconds of untimed steps before the timed region (clock state)")
    ap.add_argument("--two-in-flight", dest="two_in_flight", type=int, default=1,
                    help="rank 0, N=1: also measure the step with TWO samples in flight on two contexts (never `value`)")
    ap.add_argument("--read-sets", dest="read_sets", type=int, default=1,
                    help="rank 0, N=1, c3: also step SURVEY 8(d)'s Zipf-pool (U/N ~ 5 %%) and all-distinct (U = N) read sets (never `value`)")
    ap.add_argument("--c4-end-to-end", dest="c4_e2e", type=int, default=1,
                    help="N > 1: also time the sharded CLI's whole route, FASTQ files -> all output files, tail included (never `value`)")
    ap.add_argument("--pmc", type=int, default=1,
                    help="rank 0, N=1: measure the dominant kernel's HBM traffic with two child rocprofv3 --pmc passes")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # BASELINE.json configs[3] is what a multi-GPU run measures: 8 samples x 20 M reads, one per GPU.  A run under
    # torch.distributed.run that names no workload is that config (20 M reads per rank unless --reads says otherwise; the
    # label in config.workload carries the actual numbers); N = 1 stays C3, the config `metric` is quoted on.
    if args.workload is None:
        args.workload = "c4" if world > 1 else "c3"
    if args.reads is None:
        args.reads = 20_000_000 if args.workload == "c4" else 10_000_000
    import torch
    dist = None
    # test hooks (tests/test_gpu_parity.py runs two ranks on the one GPU of the test box):
    # MIRGE_BENCH_SHARE_GPU=1 puts every rank on device 0, MIRGE_BENCH_BACKEND=gloo keeps RCCL out of it,
    # MIRGE_BENCH_FORCE_DIST=1 takes the N > 1 route (process groups, RCCL probe, barrier, max over ranks) with ONE rank:
    # the only way RCCL itself can be brought up on a single-GPU box
    backend = os.environ.get("MIRGE_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("MIRGE_BENCH_SHARE_GPU") else local_rank
    backend_note = None
    rccl_ranks_seen = None
    if world > 1 or (os.environ.get("MIRGE_BENCH_FORCE_DIST") and "RANK" in os.environ):
        import datetime
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            # RCCL carries nothing but the barrier and one 8-byte max here.  If it cannot come up on this node (IPC mode, a
            # missing xGMI link, a driver mismatch) the measurement must not die with it: the default group is gloo (always
            # there), RCCL is a second group on top, probed with one all-reduce; the ranks agree over gloo whether every
            # one of them got it, and fall back together -- before any GPU work of the bench itself.
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
            ok = 1
            try:
                if os.environ.get("MIRGE_BENCH_FORCE_NCCL_FAIL"):  # test hook
                    raise RuntimeError("forced (MIRGE_BENCH_FORCE_NCCL_FAIL)")
                nccl_pg = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=300))
                probe = torch.ones(1, device=f"cuda:{dev_index}")
                dist.all_reduce(probe, group=nccl_pg)
                torch.cuda.synchronize()
                rccl_ranks_seen = int(probe.item())  # the probe all-reduce's sum of ones: the ranks RCCL actually connected
                ok = int(rccl_ranks_seen == world)
            except Exception as e:  # noqa: BLE001
                ok = 0
                nccl_pg = None
                backend_note = f"nccl failed on rank {rank}: {repr(e)[:200]}"
            agree = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)
            if int(agree.item()) == 1:
                bench_pg = nccl_pg
            else:
                backend, bench_pg = "gloo", None
                backend_note = backend_note or "nccl failed on another rank"
        else:
            dist.init_process_group(backend)
            bench_pg = None
    n_gpus = world if world > 1 else args.gpus
    if world == 1 and args.gpus > 1:
        print("bench.py: --gpus > 1 needs torch.distributed.run; running rank 0 only", file=sys.stderr)
        n_gpus = 1

    import __graft_entry__ as g
    if rank == 0:
        g.build()
    if dist is not None:
        dist.barrier()
    import mirge3_amd  # noqa: F401
    from mirge3_amd import _ffi, synth
    from mirge3_amd.cascade import Cascade, PASSES, EXACT_PASS, ISO_PASS

    # ---------------- inputs (not timed): libraries indexed in HBM, reads packed in HBM
    t_setup = time.perf_counter()
    sl = synth.make_libraries(seed=20260101, scale=args.scale)
    ctx = _ffi.Context(dev_index)
    # c2: exact mature-miRNA pass only; c5: exact + <=2-mismatch isomiR passes against the miRNA library,
    # then the per-position variant tally (configs[4]; BASELINE quotes it at --reads 50000000)
    n_pass = 1 if args.workload == "c2" else 9
    libs = {"mirna": sl.libs["mirna"]} if args.workload in ("c2", "c5") else sl.libs
    casc = Cascade(ctx, libs, n_pass=n_pass)
    if args.pool:
        reads = synth.make_reads(sl, args.reads, seed=1000 + rank, pool=args.pool)
    else:
        reads = synth.make_reads_chunked(sl, args.reads, seed=1000 + rank, **({} if args.long_frac is None else {"long_frac": args.long_frac}))  # one sample per rank
    raw = _ffi.DeviceReads.pack(ctx, reads)
    n_mirna = len(sl.libs["mirna"])
    t_setup = time.perf_counter() - t_setup

    state = {}

    def step():
        if args.two_calls:
            uniq = raw.collapse()
            res = casc.run(uniq)
        else:
            uniq, res = casc.collapse_and_run(raw)  # one sample: mirge_collapse_cascade
        cls, ex, iso = _ffi.count_join(ctx, uniq, res, EXACT_PASS, ISO_PASS if n_pass > ISO_PASS else -2, n_mirna)
        if args.workload == "c5":
            from mirge3_amd import a2i
            state["tally"] = a2i.tally(casc, uniq, res)["count_true"]
            if state.get("want_groups"):  # miRNA reads (rows annotated by pass 0 or 8): the tally's unit
                ps = res.fetch()[0]
                state["mirna_reads"] = int(((ps == EXACT_PASS) | (ps == ISO_PASS)).sum())
        state["U"] = len(uniq)
        state["cls"] = cls
        if state.get("want_groups"):
            state["groups"] = uniq.group_counts()
        res.close()
        uniq.close()

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier(group=bench_pg)
        torch.cuda.synchronize()

    # warm-up: builds the probe tables on first use and warms the buffer pool; its last two steps run
    # with every launch bracketed by HIP events -> the per-kernel table and the dominant kernel
    n_prof = min(2, args.warmup) if args.warmup else 1
    for _ in range(max(args.warmup - n_prof, 0)):
        step()
    ctx.profile(True)
    ctx.profile_only("")
    ctx.profile_reset()
    state["want_groups"] = True
    for _ in range(n_prof):
        step()
    state["want_groups"] = False
    recs_all = ctx.profile_records()
    table = {name: dict(launches=l, avg_ms=ms / l, total_ms=ms, units_per_launch=u / l) for name, l, ms, u in recs_all if l}
    dom = max(table, key=lambda k: table[k]["total_ms"])
    # timed region: only the dominant kernel stays bracketed (two events per launch; bracketing
    # all ~90 launches of a step costs ~0.4 ms of a ~3 ms step)
    ctx.profile_only(dom.split(".")[0] + "." if "." in dom else dom)
    # ... and its units (reads handed to every pass) are those of the profiled warm-up steps -- the same batch: reading the
    # survivor counters back is a device-to-host copy between k_resolve and the join in every step (~10 us of queue gap)
    ctx.profile_units(False)

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=f"cuda:{dev_index}" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=bench_pg)
        return float(t.item())

    # clocks up before anything is timed: W warm-up steps of ~2 ms leave a GPU that idled through the setup at whatever
    # power state it was in; untimed steps until the chip has worked for `--spinup` seconds
    # the interpreter's cyclic collector walks every container object when its oldest generation comes due -- with the
    # synthetic libraries' ~0.2 M names alive that is one ~45 ms pause every ~1000 steps, inside a 1.6 ms step; what exists
    # now is set aside (gc.freeze) so that collections during the timed region only see the steps' own few objects
    import gc
    gc.collect()
    gc.freeze()
    t_spin = time.perf_counter()
    n_spin = 0
    while time.perf_counter() - t_spin < args.spinup:
        step()
        n_spin += 1
    clocks_before = gpu_clocks_mhz()
    # timed: EXACTLY K steps per region, barrier + synchronize on both sides, max over ranks.  A region of K = 20 steps is
    # 0.04 s -- one slow launch moves it by percents -- so the region is repeated until `--min-seconds` of timed work have
    # been seen (the same number of regions on every rank); ms_per_step = all timed time / all timed steps.
    ctx.profile_reset()
    region_s, step_ms = [], []
    while True:
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ts = time.perf_counter()
            step()  # ends with the count tables on the host: synchronous
            step_ms.append((time.perf_counter() - ts) * 1e3)
        barrier()
        region_s.append(max_over_ranks(time.perf_counter() - t0))  # the same number on every rank: so is the decision below
        if sum(region_s) >= args.min_seconds or len(region_s) >= 5000:
            break
    n_regions = len(region_s)
    clocks_after = gpu_clocks_mhz()
    elapsed = float(sum(region_s))
    timed_steps = args.steps * n_regions
    recs = ctx.profile_records()
    ctx.profile(False)
    ctx.profile_only("")
    ctx.profile_units(True)
    ms_per_step = elapsed / timed_steps * 1e3
    value = n_gpus * args.reads / (elapsed / timed_steps) / 1e6
    per_region_ms = sorted(r / args.steps * 1e3 for r in region_s)
    sm = np.sort(np.asarray(step_ms))

    # ---------------- per-kernel accounting
    # `kernels` (all launches bracketed) comes from the profiled warm-up steps; the dominant kernel's
    # roofline numbers come from the HIP events recorded inside the timed region
    kernels = table
    timed = {name: dict(launches=l, avg_ms=ms / l, total_ms=ms, units_per_launch=u / l) for name, l, ms, u in recs if l}
    kd = dict(timed.get(dom) or table[dom])
    if not kd["units_per_launch"]:
        kd["units_per_launch"] = table[dom]["units_per_launch"]
    # the whole-cascade kernels are priced as SURVEY 8(d) prices them: 14 B per COLLAPSED read of their group (the profile
    # record's own units are reads handed to passes, summed over the passes: kept as `per_pass_units`)
    whole_cascade = dom.startswith("k_cascade")
    per_pass_units = kd["units_per_launch"]
    if whole_cascade:
        gi, has_n = group_index(dom)
        gc = state["groups"]
        kd["units_per_launch"] = float(gc[gi + (len(gc) // 2 if has_n else 0)])
    achieved = algo_bytes(dom) * kd["units_per_launch"] / (kd["avg_ms"] * 1e-3) / 1e9
    n_tab_steps = n_prof
    stage_ms = {
        "collapse": sum(v["total_ms"] for k, v in kernels.items() if "collapse" in k or "heads" in k or "scan" in k or "hist" in k or "k_part" in k or "flags" in k) / n_tab_steps,
        "cascade": sum(v["total_ms"] for k, v in kernels.items() if k.startswith(("k_pass", "k_resolve", "k_cascade"))) / n_tab_steps,
        "join": sum(v["total_ms"] for k, v in kernels.items() if k.startswith("k_join")) / n_tab_steps,
    }
    U = state["U"]
    out = {
        "metric": "M reads/s through full annotation cascade; per-class counts bit-exact vs ref",
        "value": round(value, 3), "unit": "M reads/s", "n_gpus": n_gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {
            "workload": {"c3": f"C3: {args.reads / 1e6:g}M-read human-like sample, collapse -> full 9-pass cascade -> count join, 1 sample per GPU",
                         "c4": f"C4: {n_gpus} samples x {args.reads / 1e6:g}M reads sharded one-per-GPU (no RCCL on the data path; BASELINE "
                               "configs[3] = 8 x 20M): every rank runs collapse -> full 9-pass cascade -> count join on its own human-like sample",
                         "c2": f"C2: {args.reads / 1e6:g}M-read human-like sample, collapse -> exact mature-miRNA pass only -> count join",
                         "c5": f"C5: {args.reads / 1e6:g}M reads, collapse -> exact + <=2-mismatch isomiR passes vs the miRNA library -> count join -> "
                               "per-position variant tally"}[args.workload],
            "raw_reads_per_gpu": args.reads, "unique_reads_per_gpu": U, "library_scale": args.scale,
            "read_templates": args.pool or None, "long_read_share": args.long_frac,
            "library_bases": {k: v.total_len for k, v in libs.items()}, "passes": n_pass,
            "sharding": f"{n_gpus} sample(s), one per GPU, no collective",
        },
        "timing": {
            "timed_steps": timed_steps, "regions": n_regions, "steps_per_region": args.steps, "timed_seconds": round(elapsed, 4),
            "region_ms_per_step": {"min": round(per_region_ms[0], 4), "median": round(per_region_ms[len(per_region_ms) // 2], 4),
                                   "max": round(per_region_ms[-1], 4)},
            "step_ms_rank0": {"min": round(float(sm[0]), 4), "p10": round(float(sm[int(0.1 * (len(sm) - 1))]), 4),
                              "median": round(float(sm[len(sm) // 2]), 4), "p90": round(float(sm[int(0.9 * (len(sm) - 1))]), 4),
                              "max": round(float(sm[-1]), 4)},
            "first_region_ms_per_step": round(region_s[0] / args.steps * 1e3, 4),
            "spinup_steps": n_spin, "gpu_sclk_mhz": {"before": clocks_before, "after": clocks_after},
            "note": "every region is exactly --steps steps between barrier + synchronize pairs, max over ranks; regions are "
                    "repeated until --min-seconds of timed work; ms_per_step / value = all regions' time / all regions' steps; "
                    "step_ms_rank0 are host-clock times of single steps (a step ends with its count tables on the host)",
        },
        "barrier_backend": (backend if dist is not None else None), "barrier_backend_note": backend_note,
        "collapsed_reads_per_s_M": round(n_gpus * U / (ms_per_step * 1e-3) / 1e6, 3),
        "collapsed_reads_note": "unique (collapsed) reads of the sample / ms_per_step: the cascade's input rate over the whole step "
                                "(north_star's >= 50 M/s target is on this unit); `value` counts raw reads",
        "rccl_ranks_seen": rccl_ranks_seen,
        "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()},
        "stage_ms_note": "sums of per-kernel HIP-event times from the profiled warm-up steps; kernels of the small read "
                         "groups overlap the big group's on streams of their own, so the sums exceed ms_per_step",
        "roofline": {
            "bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
            "algorithmic_bytes_per_unit": algo_bytes(dom), "units_per_launch": round(kd["units_per_launch"], 1),
            "unit_is": ("collapsed read of the kernel's read group (SURVEY.md 8d 'Cascade (fused minimum)': 9 B in + 5 B out); "
                        "units_per_launch = reads of that group in the sample" if whole_cascade else
                        "the record's own unit (DESIGN.md section 4 table)"),
            "avg_launch_ms": round(kd["avg_ms"], 5), "launches": kd["launches"],
            "note": "integer / indexing kernel bound by random 64-B sectors and vector issue, not by streaming bandwidth; HBM "
                    "fraction on SURVEY 8(d)'s algorithmic bytes as the brief requires; timed with HIP events on its launch stream "
                    "inside the timed region"
                    + ("; k_cascade_bulk = ALL passes of the bulk read group in one launch (MIRGE_BULK_FUSED=0: one launch per pass)"
                       if dom.startswith("k_cascade_bulk") else ""),
        },
        "kernels": {k: {"launches": v["launches"], "avg_ms": round(v["avg_ms"], 5),
                        "units_per_launch": round(v["units_per_launch"], 1)} for k, v in sorted(kernels.items())},
        "setup_s": round(t_setup, 1),
    }
    if whole_cascade and per_pass_units:
        a18 = algo_bytes(dom, per_pass=True) * per_pass_units / (kd["avg_ms"] * 1e-3) / 1e9
        out["roofline"]["per_pass_units"] = {
            "units_per_launch": round(per_pass_units, 1), "algorithmic_bytes_per_unit": algo_bytes(dom, per_pass=True),
            "achieved": round(a18, 3), "frac": round(a18 / HBM_PEAK_GBS, 6), "unit": "GB/s",
            "note": "the same launch priced per (read, pass): a unit is a read handed to a pass (k_pass's 4 B index + 9 B read in, "
                    "5 B out), summed over the passes from the per-workgroup survivor counters of the bracketed warm-up steps; "
                    "rounds 1-3 reported this figure as `frac`"}

    if rank == 0:
        try:
            out["roofline_stages"] = stage_rooflines(kernels, n_tab_steps, raw.group_counts(), state["groups"], 1, n_mirna, out["roofline"],
                                                     state.get("mirna_reads") if args.workload == "c5" else None)
            out["roofline_stages_note"] = ("every stage of the step on SURVEY 8(d)'s algorithmic bytes against the 8 TB/s HBM peak; times = HIP "
                                           "events of the bracketed warm-up steps, bulk read group's kernels of the stage (its critical path)")
            w = _ffi.cascade_walks(ctx)
            out["cascade_walks"] = {"walks": w[0], "exact_lookup_passes": w[1], "passes": w[2],
                                    "note": "bulk group of one-word reads: walks over its survivor lists; passes answered by one whole-read "
                                            "lookup inside another pass's walk (round 5: exact miRNA, primary tRNA); merged runs count once"}
        except Exception as e:  # noqa: BLE001 -- a secondary figure must not take the line down
            out["roofline_stages"] = {"error": repr(e)[:300]}

    # ---------------- SURVEY 8(d)'s two read sets beside the default draw (never `value`): Zipf duplication over a template pool
    # (U/N of a few percent, the "realistic" set) and all-distinct reads (U = N, "stress").  The default sample -- every read
    # drawn independently, U/N = 0.42 -- is neither; `value` swings 2 x between the two, so the line carries all three.
    if rank == 0 and n_gpus == 1 and args.read_sets and args.workload == "c3" and not args.pool:
        try:
            out["read_sets"] = secondary_read_sets(args, sl, ctx, casc, _ffi, synth, n_mirna, EXACT_PASS, ISO_PASS, out)
        except Exception as e:  # noqa: BLE001
            out["read_sets"] = {"error": repr(e)[:300]}

    if rank == 0 and n_gpus == 1 and args.read_sets and args.workload == "c3" and not args.pool:
        try:
            out.setdefault("read_sets", {})["repeat_rich"] = repeat_rich_leg(args, ctx, _ffi, synth, Cascade, PASSES, n_mirna, EXACT_PASS, ISO_PASS,
                                                                              out["ms_per_step"])
        except Exception as e:  # noqa: BLE001
            out.setdefault("read_sets", {})["repeat_rich"] = {"error": repr(e)[:300]}

    # ---------------- the same from the FILE's text (never `value`): FASTQ bytes in host memory -> records parsed,
    # filtered, packed on the GPU -> collapse -> cascade -> count tables on the host
    if rank == 0 and n_gpus == 1 and len(reads) <= 20_000_000:  # the text is built with numpy: ~0.5 KB of host memory per read
        text = fastq_text(reads)
        best_dt = None
        for _ in range(3):
            t = time.perf_counter()
            r_t, n_rec = _ffi.DeviceReads.parse(ctx, text, 1, 16)
            u_t = r_t.collapse()
            res_t = casc.run(u_t)
            cls_t, _, _ = _ffi.count_join(ctx, u_t, res_t, EXACT_PASS, ISO_PASS if n_pass > ISO_PASS else -2, n_mirna)
            dt = time.perf_counter() - t
            best_dt = dt if best_dt is None else min(best_dt, dt)
            ok_t = n_rec == len(reads) and len(u_t) == state["U"] and np.array_equal(cls_t, state["cls"])
            res_t.close(); u_t.close(); r_t.close()
        out["raw_reads_per_s_fastq_in_M"] = round(args.reads / best_dt / 1e6, 2)  # SURVEY 8(d) "raw reads/s": FASTQ records in -> counts out
        out["fastq_text_path"] = {"M_reads_per_s": round(args.reads / best_dt / 1e6, 2), "ms": round(best_dt * 1e3, 2),
                                  "text_MB": round(text.size / 1e6, 1), "same_counts_as_step": bool(ok_t),
                                  "note": "FASTQ text (4-line records, pageable host memory) over PCIe, mirge_reads_parse on the "
                                          "GPU, collapse, cascade, count tables back; best of 3 passes, not part of `value`"}
        # ---------------- the same with the reads as a sequencer delivers them (never `value`): insert + 3' adapter, 50 cycles,
        # trimmed on the GPU (mirge_reads_parse_trim: quality trimming -q 10 + adapter removal, the reference's `-a illumina`)
        if args.workload == "c3" and len(reads) <= 10_000_000:
            from mirge3_amd.collapse import ILLUMINA_3P
            ad = np.frombuffer(ILLUMINA_3P.encode(), dtype=np.uint8)
            L = reads.lengths
            L2 = np.minimum(L + ad.shape[0], 50)
            off2 = np.zeros(len(reads) + 1, dtype=np.int64)
            np.cumsum(L2, out=off2[1:])
            data2 = np.empty(int(off2[-1]), dtype=np.uint8)
            rows = np.repeat(np.arange(len(reads), dtype=np.int64), L2)
            within = np.arange(int(off2[-1]), dtype=np.int64) - off2[:-1][rows]
            ins = within < L[rows]
            data2[ins] = reads.data[(reads.offsets[:-1][rows] + within)[ins]]
            data2[~ins] = ad[(within - L[rows])[~ins]]
            del rows, within, ins
            from mirge3_amd.seqio import FlatSeqs as _FS
            text2 = fastq_text(_FS(data2, off2))
            trim = _ffi.MirgeTrim.make(adapter=ILLUMINA_3P, quality_back=10, count_per_modifier=False)
            best_dt = best_parse = None
            for _ in range(3):
                t = time.perf_counter()
                r_t, n_rec = _ffi.DeviceReads.parse(ctx, text2, 1, 16, trim)
                t_parse = time.perf_counter() - t
                u_t = r_t.collapse()
                res_t = casc.run(u_t)
                cls_t, _, _ = _ffi.count_join(ctx, u_t, res_t, EXACT_PASS, ISO_PASS if n_pass > ISO_PASS else -2, n_mirna)
                dt = time.perf_counter() - t
                if best_dt is None or dt < best_dt:
                    best_dt, best_parse = dt, t_parse
                n_trimmed, u_trimmed = len(r_t), len(u_t)
                res_t.close(); u_t.close(); r_t.close()
            # the same text through k_trim's GENERAL instance (what two adapters, -n, --no-indels, wildcards in the read, anchored and
            # linked adapters run): here only --match-read-wildcards set, so the reads kept are the same
            gen_parse = None
            try:
                trim_g = _ffi.MirgeTrim.make(adapter=ILLUMINA_3P, quality_back=10, count_per_modifier=False, read_wildcards=True)
                for _ in range(2):
                    t = time.perf_counter()
                    r_g, _n = _ffi.DeviceReads.parse(ctx, text2, 1, 16, trim_g)
                    dtg = time.perf_counter() - t
                    gen_parse = dtg if gen_parse is None else min(gen_parse, dtg)
                    n_general = len(r_g)
                    r_g.close()
            except Exception as e:  # noqa: BLE001
                gen_parse, n_general = None, repr(e)[:200]
            out["fastq_trim_path"] = {"M_reads_per_s": round(args.reads / best_dt / 1e6, 2), "ms": round(best_dt * 1e3, 2),
                                      "parse_trim_ms": round(best_parse * 1e3, 2), "text_MB": round(text2.size / 1e6, 1),
                                      "parse_trim_general_instance_ms": None if gen_parse is None else round(gen_parse * 1e3, 2),
                                      "reads_kept_general_instance": n_general,
                                      "reads_kept": int(n_trimmed), "unique": int(u_trimmed),
                                      "note": "inserts + TruSeq small-RNA 3' adapter cut at 50 cycles; quality trimming + adapter removal "
                                              "(k_trim) + parse + collapse + cascade + count tables; the adapter's first bases also occur in "
                                              "inserts by chance, so a few reads come out shorter than their insert, as with cutadapt; "
                                              "best of 3, not part of `value`"}
            del text2, data2
        # ---------------- the CLI's own route (never `value`): FASTQ FILE on disk -> every output file of the hot path
        # (miR.Counts.csv, miR.RPM.csv, annotation.report.csv/html, mapped.csv, unmapped.csv), mirge3_amd.fastpath.run
        if args.cli_path:
            try:
                out["cli_path"] = cli_path(args, sl, libs, text, n_pass)
            except Exception as e:
                out["cli_path"] = {"error": repr(e)[:300]}
        del text

    # ---------------- BASELINE configs[3] end to end (never `value`): all ranks, N > 1 -- the product's route with its tail
    if dist is not None and world > 1 and args.c4_e2e and args.workload in ("c3", "c4") and not args.pool:
        try:
            e2e = c4_end_to_end(args, dist, rank, world, sl, casc, reads, n_pass)
            if rank == 0 and e2e is not None:
                out["c4_end_to_end"] = e2e
        except Exception as e:  # noqa: BLE001 -- a secondary leg must not take the line down
            if rank == 0:
                out["c4_end_to_end"] = {"error": repr(e)[:400]}

    # ---------------- two samples in flight (never `value`): what a batch of samples per GPU runs at -- a second context (its own
    # streams, its own copy of the libraries) steps a second sample from a second host thread; the collapse of one sample (LDS-
    # and write-bound) overlaps the cascade of the other (bound by L1 misses in flight)
    if rank == 0 and n_gpus == 1 and args.two_in_flight and args.workload in ("c3", "c4") and not args.pool:
        try:
            import threading
            ctx2 = _ffi.Context(dev_index)
            casc2 = Cascade(ctx2, libs, n_pass=n_pass)
            raw2 = _ffi.DeviceReads.pack(ctx2, reads)

            def step2():
                u2, r2 = casc2.collapse_and_run(raw2)
                _ffi.count_join(ctx2, u2, r2, EXACT_PASS, ISO_PASS if n_pass > ISO_PASS else -2, n_mirna)
                r2.close(); u2.close()

            for _ in range(3):
                step2()
            K2 = max(20, min(200, int(0.25 / max(ms_per_step * 1e-3, 1e-6))))
            t = time.perf_counter()
            for _ in range(K2):
                step()
            one = (time.perf_counter() - t) / K2
            th = [threading.Thread(target=lambda f=f: [f() for _ in range(K2)]) for f in (step, step2)]
            t = time.perf_counter()
            for x in th:
                x.start()
            for x in th:
                x.join()
            two = (time.perf_counter() - t) / (2 * K2)
            out["two_samples_in_flight"] = {
                "M_reads_per_s": round(args.reads / two / 1e6, 1), "ms_per_sample": round(two * 1e3, 4),
                "one_at_a_time_ms": round(one * 1e3, 4), "speedup": round(one / two, 3), "steps_each": K2,
                "note": "two contexts (two sets of HIP streams, two host threads) step the same workload side by side on the one "
                        "GPU: what a batch of samples per GPU would gain from it -- nothing since round 3 (a single sample's step "
                        "already runs on five streams; round 2 measured 1.09-1.14 x); `value` is one sample at a time"}
            raw2.close(); casc2.close(); ctx2.close()
        except Exception as e:  # noqa: BLE001
            out["two_samples_in_flight"] = {"error": repr(e)[:300]}

    # ---------------- SURVEY 8(d)'s "collapsed reads/s" to the letter (never `value`): U device-resident packed unique reads ->
    # cascade -> count join -> per-read annotation (pass, reference, offset, mismatches: 10 B per read) AND the count tables back
    # on the host.  `collapsed_reads_per_s_M` above divides U by the whole step (collapse included, no per-read fetch).
    if rank == 0:
        try:
            u_c = raw.collapse()
            best_dt = None
            for _ in range(4):  # best of four: the first pass also pays for new buffer-pool blocks
                ctx.sync()
                t = time.perf_counter()
                res_c = casc.run(u_c)
                ann_c = res_c.fetch()
                _ffi.count_join(ctx, u_c, res_c, EXACT_PASS, ISO_PASS if n_pass > ISO_PASS else -2, n_mirna)
                dt = time.perf_counter() - t
                best_dt = dt if best_dt is None else min(best_dt, dt)
                res_c.close()
            out["collapsed_reads_annotation_on_host"] = {
                "M_collapsed_reads_per_s": round(len(u_c) / best_dt / 1e6, 1), "ms": round(best_dt * 1e3, 3), "unique_reads": len(u_c),
                "annotated": int((ann_c[0] >= 0).sum()),
                "note": "SURVEY 8(d)'s second rate as written: the sample's collapsed reads resident in HBM -> cascade -> count join -> "
                        "per-read annotation (10 B per read over PCIe) + count tables on the host; best of 4, not part of `value`; "
                        "north_star's >= 50 M collapsed reads/s target is on this unit"}
            u_c.close()
        except Exception as e:  # noqa: BLE001
            out["collapsed_reads_annotation_on_host"] = {"error": repr(e)[:300]}

    # ---------------- SURVEY 8(f) row N2 (never `value`): the isomiR typing of the miRTop GFF3 on the sample's miRNA reads --
    # every read of the exact-miRNA and isomiR passes through k_isotype (the reference's per-read difflib diff + list rewriting,
    # summary.py:204-470), records back on the host; the kernel's own time from its HIP-event bracket
    if rank == 0 and n_gpus == 1 and args.workload in ("c3", "c5") and n_pass > ISO_PASS and hasattr(sl, "mir_hairpin"):
        try:
            from mirge3_amd import gff
            mir, hp = sl.libs["mirna"], sl.libs["hairpin"]
            mseq, hseq = mir.seqs.to_list(), hp.seqs.to_list()
            pre_of = {nm: hp.names[int(sl.mir_hairpin[k])] for k, nm in enumerate(mir.names)}
            tabs = gff.resolve_names(mir.names, dict(zip(mir.names, mseq)), pre_of, dict(zip(hp.names, hseq)))
            u_g = raw.collapse()
            res_g = casc.run(u_g)
            ps_g = res_g.fetch()[0]
            rows_g = np.nonzero((ps_g == EXACT_PASS) | (ps_g == ISO_PASS))[0].astype(np.int64)
            ctx.profile(True); ctx.profile_only("k_isotype"); ctx.profile_reset()
            best_dt = None
            for _ in range(3):
                ctx.sync()
                t = time.perf_counter()
                recs_g = gff.isomir_records(casc, u_g, res_g, tabs, rows_g)
                dt = time.perf_counter() - t
                best_dt = dt if best_dt is None else min(best_dt, dt)
            k_l, k_ms = 0, 0.0
            for name, l, ms, u in ctx.profile_records():
                if name.startswith("k_isotype") and l:
                    k_l, k_ms = k_l + l, k_ms + ms
            ctx.profile(False); ctx.profile_only("")
            k_ms_call = k_ms / 3.0
            # round 6: the whole file from the device (rows chosen, typed, measured, formatted there; the text over PCIe; positional writes)
            dev_file = None
            try:
                import tempfile
                order_g = u_g.first_appearance_order()
                with tempfile.TemporaryDirectory(prefix="mirge_gff_", dir="/tmp") as td:
                    best_f = None
                    for _ in range(3):
                        ctx.sync()
                        t = time.perf_counter()
                        o_g = gff.write_gff_device_with(tabs, os.path.join(td, "sample_miRge3.gff"), "miRBase", ["S1"], casc, u_g, res_g, order_g)
                        dtf = time.perf_counter() - t
                        best_f = dtf if best_f is None else min(best_f, dtf)
                    dev_file = {"ms_call": round(best_f * 1e3, 3), "lines": o_g["lines"],
                                "file_MB": round(os.path.getsize(os.path.join(td, "sample_miRge3.gff")) / 1e6, 1),
                                "note": "mirge_gff_write_device: frame order in, sample_miRge3.gff on disk out (best of 3); nothing per read reaches "
                                        "the host but the file's text"}
            except Exception as e:  # noqa: BLE001
                dev_file = {"error": repr(e)[:300]}
            b = ALGO_BYTES["k_isotype"] * float(rows_g.shape[0])
            out["gff_typing"] = {
                "mirna_reads": int(rows_g.shape[0]), "isomir_records": int((recs_g["kind"] == 2).sum()),
                "ms_call": round(best_dt * 1e3, 3), "M_reads_per_s_call": round(rows_g.shape[0] / best_dt / 1e6, 1),
                "k_isotype_ms": round(k_ms_call, 4), "k_isotype_launches_per_call": round(k_l / 3.0, 1), "file_from_the_device": dev_file,
                "roofline": {"bound": "hbm", "algorithmic_bytes": b, "achieved": round(b / max(k_ms_call, 1e-9) / 1e6, 2), "unit": "GB/s",
                             "peak": HBM_PEAK_GBS, "frac": round(b / max(k_ms_call, 1e-9) / 1e6 / HBM_PEAK_GBS, 6),
                             "bytes_per_unit": "14 in + one 336-byte record out per miRNA read"},
                "note": "mirge_isomir_type on every read the exact-miRNA / isomiR passes annotated: slot map up, kernel, 336-byte records "
                        "back over PCIe (ms_call, best of 3); k_isotype_ms = the kernel(s) alone by HIP events; the file itself "
                        "(mirge_gff_write, host cores) is part of cli_path when -gff is given, not of this leg"}
            res_g.close(); u_g.close()
        except Exception as e:  # noqa: BLE001
            out["gff_typing"] = {"error": repr(e)[:300]}

    # ---------------- PCIe-inclusive rate (never `value`): host ASCII reads in, per-read annotation + counts out
    if rank == 0:
        best_dt = None
        for _ in range(3):  # best of three: the first pass also pays for new buffer-pool blocks
            t = time.perf_counter()
            r_h = _ffi.DeviceReads.pack(ctx, reads)
            u_h = r_h.collapse()
            res_h = casc.run(u_h)
            ann = res_h.fetch()
            cnt_h, first_h = u_h.counts()
            _ffi.count_join(ctx, u_h, res_h, EXACT_PASS, ISO_PASS if n_pass > ISO_PASS else -2, n_mirna)
            dt = time.perf_counter() - t
            best_dt = dt if best_dt is None else min(best_dt, dt)
            res_h.close(); u_h.close(); r_h.close()
        out["host_buffer_path"] = {"M_reads_per_s": round(args.reads / best_dt / 1e6, 2), "ms": round(best_dt * 1e3, 2),
                                   "note": "ASCII reads + offsets over PCIe, 2-bit pack on the GPU, annotation (10 B/unique) "
                                           "and counts back to numpy; best of 3 passes, not part of `value`"}

    # ---------------- CPU baseline (rank 0, N = 1): the reference path (bowtie 1.x, the ten argument strings of
    # manifoldAlign.py:85 verbatim) when the box has one, else the oracle; on a bounded sample, and parity on it
    out["parity"] = ("GPU path == in-repo oracle (restated bowtie-1 manual semantics) and == the files the reference's own "
                     "bwtAlign/summarize wrote around that oracle; the bowtie predicate itself is UNPINNED (no bowtie 1.x "
                     "in the image or on the pool): 'bit-exact vs ref' in `metric` is BASELINE.json's wording, not a claim")
    bowtie_dir = reference_bowtie_dir()
    if rank == 0 and n_gpus == 1 and args.cpu_baseline and bowtie_dir:
        try:
            out["cpu_baseline"] = reference_bowtie_baseline(bowtie_dir, sl, libs, reads, n_pass, ctx, casc, args)
            out["parity"] = out["cpu_baseline"].pop("parity")
        except Exception as e:  # a broken bowtie install must not take the bench line down
            out["cpu_baseline_reference_error"] = repr(e)[:300]
    if rank == 0 and n_gpus == 1 and args.cpu_baseline and "cpu_baseline" not in out:
        import oracle
        cores = os.cpu_count() or 1
        threads = min(cores, 64)
        libs_o = [(sl.libs[PASSES[p][1]].seqs.data, sl.libs[PASSES[p][1]].seqs.offsets) for p in range(n_pass)]

        def cpu_run(m):
            oracle.build_seconds(reset=True)
            t = time.perf_counter()
            sub_off = reads.offsets[: m + 1]
            sub_data = reads.data[: sub_off[-1]]
            first, cnt, _ = oracle.collapse(sub_data, sub_off)
            from mirge3_amd.seqio import FlatSeqs
            u = FlatSeqs(sub_data, sub_off).take(first)
            ps, ref, off, mm = oracle.cascade(u.data, u.offsets, libs_o, n_pass=n_pass, indexed=True, threads=threads)
            cls = np.array([cnt[ps == p].sum() for p in range(n_pass)], dtype=np.int64)
            wall = time.perf_counter() - t
            return wall, wall - oracle.build_seconds(), cls, u

        m2 = max(1000, min(args.cpu_sample, args.reads))
        wall, work, cls_o, u2 = cpu_run(m2)
        out["cpu_baseline"] = {
            "value": round(m2 / max(work, 1e-9) / 1e6, 4), "unit": "M reads/s", "cores": threads, "kind": "port",
            "sample": f"oracle (C restatement, OpenMP {threads} threads) on the first {m2} raw reads of the same sample: "
                      f"collapse + 9-pass cascade + class sums, {work:.2f} s; its k-mer table construction "
                      f"({wall - work:.2f} s, the oracle's 'bowtie-build') is excluded",
        }
        # live parity on the sample: per-class counts, GPU vs oracle
        sub = reads.take(np.arange(m2))
        r2 = _ffi.DeviceReads.pack(ctx, sub)
        u_g = r2.collapse()
        res = casc.run(u_g)
        cls_g, _, _ = _ffi.count_join(ctx, u_g, res, EXACT_PASS, ISO_PASS if n_pass > ISO_PASS else -2, n_mirna)
        out["parity_on_cpu_sample"] = bool(np.array_equal(cls_g[:, 0], cls_o)) and len(u_g) == len(u2)
        res.close(); u_g.close(); r2.close()

    if rank == 0 and n_gpus == 1 and args.pmc:
        t = pmc_traffic(args, dom)
        if t is not None:
            algo_launch = algo_bytes(dom) * kd["units_per_launch"]
            out["roofline"]["traffic"] = round(t["bytes"], 1)
            out["roofline"]["traffic_raw"] = round(t["bytes_raw"], 1)
            out["roofline"]["traffic_detail"] = {
                "unit": "bytes per launch", "FETCH_SIZE_kB_raw": round(t["fetch_kb_raw"], 1),
                "WRITE_SIZE_kB": round(t["write_kb"], 1),
                "formula": "traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B; the x2 is "
                           "calibrated for wide streaming reads, so an upper bound for this kernel's 8/16-B random reads); "
                           "traffic_raw = (FETCH_SIZE + WRITE_SIZE)*1024 (every request taken as one 64-B sector: the lower bound)",
                "algorithmic_bytes_per_launch": round(algo_launch, 1),
                "traffic_over_algorithmic": round(t["bytes"] / max(algo_launch, 1.0), 2),
                "traffic_raw_over_algorithmic": round(t["bytes_raw"] / max(algo_launch, 1.0), 2),
            }
            # the resource this kernel actually consumes: L2-missing 64-B sectors of random table reads
            sectors = t["fetch_kb_raw"] * 1024.0 / 64.0
            rate = sectors / (out["roofline"]["avg_launch_ms"] * 1e-3) / 1e9
            out["roofline"]["random_sector"] = {
                "sectors_per_launch": round(sectors, 1), "achieved": round(rate, 2), "peak": RANDOM_SECTOR_PEAK_G,
                "unit": "G sectors/s", "frac": round(rate / RANDOM_SECTOR_PEAK_G, 4),
                "note": "FETCH_SIZE raw (one 64-B request per random read) / launch time; peak = independent random "
                        "4-B loads from a 0.25-16 GB table on this chip, tools/microbench_random.hip "
                        "(profiles/r01_microbench_random.txt)"}
            if t.get("valu_insts"):
                # SURVEY 8(d)'s second bound: vector-instruction issue.  A wave64 VALU instruction occupies its SIMD16 for 4
                # cycles, so the chip issues at most CUs x 4 SIMDs x sclk / 4 of them per second.
                # at the chip's peak shader clock: sysfs lists eight cards on these boxes without saying which one this process
                # was given (another tenant's idle card read 642 MHz and made the fraction 1.76), and the SMI samples taken
                # during a run show the busy card at 2.39-2.41 GHz
                sclk = SCLK_PEAK_MHZ
                peak = N_CU * SIMD_PER_CU * sclk * 1e6 / VALU_CYCLES_PER_WAVE64 / 1e9
                ach = t["valu_insts"] / (out["roofline"]["avg_launch_ms"] * 1e-3) / 1e9
                out["roofline"]["valu"] = {
                    "insts_per_launch": round(t["valu_insts"], 1), "achieved": round(ach, 2), "peak": round(peak, 1),
                    "unit": "G wave-level VALU instructions/s", "frac": round(ach / peak, 4), "sclk_mhz": sclk,
                    "note": "SQ_INSTS_VALU (own --pmc pass) x 4 issue cycles / (256 CUs x 4 SIMDs x 2.4 GHz peak shader clock x launch "
                            "time): the share of the chip's vector-issue cycles the kernel uses"}
    if rank == 0:
        print(json.dumps(out))
    raw.close()
    casc.close()
    if dist is not None:
        try:
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001 -- the line is out; a group a secondary leg left in a timed-out state must not turn rc != 0
            pass


if __name__ == "__main__":
    main()
