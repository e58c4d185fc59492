#!/usr/bin/env python3
"""Where a sharded CLI run's time goes at BASELINE configs[3]'s size (8 samples x 20 M reads, one per GPU) -- measured on ONE
GPU that all ranks share (MIRGE_SHARE_GPU=1 over gloo: the test pool has single-GPU boxes), so per-rank sample seconds are
upper bounds (eight processes time-share the device) while rank 0's serial tail -- what the other seven GPUs would idle
through -- is what it is on a real node: loading the ranks' dictionaries, the weighted collapse into the sample matrix, the
cascade over the joint table, mapped.csv / unmapped.csv of the union.

  python tools/sharded_c4.py [--ranks 8] [--reads 20000000] [--scale full] [--out profiles/r05_sharded_c4.txt] [--one-process 0]

Writes the table the CLI logs in run.log ("sharded run timing: {...}") plus file sizes; with --one-process 1 also runs the same
samples through ONE process (the files must be identical, byte for byte)."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--reads", type=int, default=20_000_000)
    ap.add_argument("--scale", default="full")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sharded_c4.txt"))
    ap.add_argument("--one-process", dest="one_process", type=int, default=0)
    ap.add_argument("--port", type=int, default=29571)
    ap.add_argument("--tmp", default="/tmp")
    ap.add_argument("--tails", default="ranges,rank0", help="which tails to run: ranges (round 6: every rank its key range), rank0 (round 5)")
    args = ap.parse_args()
    import numpy as np  # noqa: F401
    import mirge3_amd  # noqa: F401
    from mirge3_amd import synth
    from mirge3_amd.seqio import index_basename, write_fasta
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_tool", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    tmp = tempfile.mkdtemp(prefix="mirge_c4_", dir=args.tmp)
    report = []

    def say(line=""):
        print(line, flush=True)
        report.append(line)
    try:
        t0 = time.perf_counter()
        sl = synth.make_libraries(seed=20260101, scale=args.scale)
        idx = os.path.join(tmp, "Libs", "bench", "index.Libs")
        os.makedirs(idx)
        os.makedirs(os.path.join(tmp, "Libs", "bench", "annotation.Libs"))
        for key, lib in sl.libs.items():
            write_fasta(os.path.join(idx, index_basename("bench", key, "miRBase") + ".fa"), lib)
        with open(os.path.join(tmp, "Libs", "bench", "annotation.Libs", "bench_merges_miRBase.csv"), "w") as fh:
            fh.write("".join(",".join(r) + "\n" for r in sl.merges))
        files = []
        for i in range(args.ranks):
            reads = synth.make_reads_chunked(sl, args.reads, seed=1000 + i)
            fq = os.path.join(tmp, f"S{i}.fastq")
            bench.fastq_text(reads).tofile(fq)
            files.append(fq)
            del reads
        say(f"# {args.ranks} samples x {args.reads / 1e6:g} M reads ({args.scale} libraries), FASTQ files of "
            f"{os.path.getsize(files[0]) / 1e6:.0f} MB each, written in {time.perf_counter() - t0:.0f} s")
        launcher = os.path.join(tmp, "run_cli.py")
        with open(launcher, "w") as fh:
            fh.write("import sys; sys.path.insert(0, %r); import mirge3_amd; from mirge3_amd.cli import main; main()\n" % ROOT)
        base = ["-s", ",".join(files), "-lib", os.path.join(tmp, "Libs"), "-on", "bench", "-db", "miRBase", "-o", tmp, "-shh"]
        env = dict(os.environ, MIRGE_SHARE_GPU="1", MIRGE_LIB_CACHE="1", OMP_NUM_THREADS=str(max(1, (os.cpu_count() or 8) // args.ranks)))
        # a warm-up run of one small sample writes the library cache (what a production node has next to its indexes)
        subprocess.run([sys.executable, launcher, "-s", files[0], "-lib", os.path.join(tmp, "Libs"), "-on", "bench", "-db", "miRBase", "-o", tmp,
                        "-shh", "-dn", "warm"], env=env, check=True, capture_output=True, text=True, timeout=1800)
        shutil.rmtree(os.path.join(tmp, "warm"), ignore_errors=True)
        tails = [x for x in args.tails.split(",") if x]
        for ti, tail in enumerate(tails):
            dn = "sharded" if ti == 0 else "sharded_" + tail
            t = time.perf_counter()
            r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.ranks), "--master-addr",
                                "127.0.0.1", "--master-port", str(args.port + ti), launcher] + base + ["-dn", dn],
                               env=dict(env, MIRGE_SHARD_TAIL=tail), capture_output=True, text=True, timeout=3600)
            wall = time.perf_counter() - t
            if r.returncode != 0:
                say(f"sharded run ({tail}) FAILED:\n" + r.stderr[-3000:])
                return 1
            log = open(os.path.join(tmp, dn, "run.log")).read()
            line = [ln for ln in log.splitlines() if ln.startswith("sharded run timing: ")][-1]
            d = json.loads(line[len("sharded run timing: "):])
            say(f"sharded run, tail = {d.get('tail', tail)}, {args.ranks} ranks on one GPU: {wall:.2f} s wall (process start-up, libraries from their "
                "cache and probe tables included)")
            say(f"  ranks' samples + gather: {d['samples_and_gather_s']:.3f} s")
            say("  per sample (its rank's own clock; the ranks share the GPU here): "
                + ", ".join(f"{p['name']} {p.get('sample_s', 0):.2f} s (hand-over {p.get('handover_s', 0):.2f}, U {p.get('unique_reads', 0) / 1e6:.2f} M)" for p in d["per_sample"]))
            say("  rank 0's tail:")
            for k, v in d["rank0_tail"].items():
                say(f"    {k:34s} {v}")
            if d.get("ranges_tail_per_rank"):
                say("  every rank's range tail (its own clock; on one shared GPU the device steps of the ranks queue behind each other):")
                keys = [k for k in d["ranges_tail_per_rank"][0] if k != "joint_unique_reads"]
                say("    " + " ".join(f"{k[:22]:>22s}" for k in ["rank"] + keys))
                for q, row in enumerate(d["ranges_tail_per_rank"]):
                    say("    " + " ".join(f"{str(v)[:22]:>22s}" for v in [q] + [row.get(k) for k in keys]))
            say(f"  rank 0 peak host memory: {d['rank0_peak_rss_MB']:.0f} MB")
            for f in ("mapped.csv", "unmapped.csv", "miR.Counts.csv"):
                say(f"  {f}: {os.path.getsize(os.path.join(tmp, dn, f)) / 1e6:.1f} MB")
            slowest = max(p.get("sample_s", 0) for p in d["per_sample"])
            say(f"  rank 0's tail {d['rank0_tail'].get('rank0_tail_s', 0):.2f} s against the slowest rank's sample {slowest:.2f} s")
            if ti > 0:
                for f in ("mapped.csv", "unmapped.csv", "miR.Counts.csv", "miR.RPM.csv", "annotation.report.csv"):
                    a, b = os.path.join(tmp, "sharded", f), os.path.join(tmp, dn, f)
                    same = os.path.getsize(a) == os.path.getsize(b) and subprocess.run(["cmp", "-s", a, b]).returncode == 0
                    say(f"  {f}: {'identical to the first tail' if same else 'DIFFERS from the first tail'}")
                shutil.rmtree(os.path.join(tmp, dn), ignore_errors=True)
        if args.one_process:
            t = time.perf_counter()
            r1 = subprocess.run([sys.executable, launcher] + base + ["-dn", "one"], env=env, capture_output=True, text=True, timeout=3600)
            say(f"one process, the same samples: {time.perf_counter() - t:.2f} s wall" + ("" if r1.returncode == 0 else " FAILED " + r1.stderr[-500:]))
            if r1.returncode == 0:
                for f in ("mapped.csv", "unmapped.csv", "miR.Counts.csv", "miR.RPM.csv", "annotation.report.csv"):
                    a, b = os.path.join(tmp, "one", f), os.path.join(tmp, "sharded", f)
                    same = os.path.getsize(a) == os.path.getsize(b) and subprocess.run(["cmp", "-s", a, b]).returncode == 0
                    say(f"  {f}: {'identical' if same else 'DIFFERS'}")
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as fh:
            fh.write("\n".join(report) + "\n")
        return 0
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
