"""Interleaved A/B of several builds / environments of libmirge_native.so, each bench run in a fresh process.
usage: python tools/ab_multi.py [--rounds N] [--bench-args "..."] name=[path.so][,ENV=VALUE...] ...
  (an empty path = the in-tree library).  Prints ms per step and the average launch time of the step's main kernels per run,
  then the minimum and median per configuration."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds, extra = 3, []
while args and args[0].startswith("--"):
    if args[0] == "--rounds":
        rounds = int(args[1]); args = args[2:]
    elif args[0] == "--bench-args":
        extra = args[1].split(); args = args[2:]
    else:
        raise SystemExit(__doc__)
cfgs = []
for a in args:
    name, _, rest = a.partition("=")
    parts = rest.split(",") if rest else [""]
    so = parts[0]
    env = dict(p.split("=", 1) for p in parts[1:] if p)
    cfgs.append((name, os.path.abspath(so) if so else "", env))
KEYS = ("k_part_agg.w1", "k_part_split.w1", "k_part_dedup.w1", "k_cascade_bulk.w1", "k_resolve.w1", "k_cascade_fused.w2", "k_part_compact.w1")
res = {c[0]: [] for c in cfgs}
ker = {c[0]: {k: [] for k in KEYS} for c in cfgs}
for r in range(rounds):
    for name, so, envx in cfgs:
        env = dict(os.environ)
        env.pop("MIRGE_NATIVE_SO", None)
        if so:
            env["MIRGE_NATIVE_SO"] = so
        env.update(envx)
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "30", "--warmup", "3", "--cpu-baseline", "0", "--pmc", "0",
                              "--two-in-flight", "0", "--cli-path", "0", "--read-sets", "0"] + extra, env=env, capture_output=True, text=True)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not lines:
            print(name, "FAILED", out.stderr[-400:], flush=True)
            continue
        d = json.loads(lines[-1])
        res[name].append(d["ms_per_step"])
        ks = {k: round(d["kernels"][k]["avg_ms"], 4) for k in KEYS if k in d.get("kernels", {})}
        for k, v in ks.items():
            ker[name][k].append(v)
        print(f"{name:14s} {d['ms_per_step']:.4f}", ks, flush=True)
print()
for name, v in res.items():
    if v:
        med = lambda x: sorted(x)[len(x) // 2]
        print(f"{name:14s} min {min(v):.4f} median {med(v):.4f} | " + "  ".join(f"{k[2:-3]} {med(x):.4f}" for k, x in ker[name].items() if x))
