"""Interleaved A/B of walk groupings of k_cascade_bulk (MIRGE_WALKS) and, optionally, of another build of the library.
usage: python tools/ab_walks.py [rounds] [variant.so]     -> ms per step and ms per k_cascade_bulk launch for every setting"""
import json, os, subprocess, sys
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
var = os.path.abspath(sys.argv[2]) if len(sys.argv) > 2 else None
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
settings = [("default", {}), ("1x7", {"MIRGE_WALKS": "1,1,1,1,1,1,1"}), ("4,1,1,1", {"MIRGE_WALKS": "4,1,1,1"}), ("2,2,2,1", {"MIRGE_WALKS": "2,2,2,1"}),
            ("4,3", {"MIRGE_WALKS": "4,3"}), ("2,2,1,1,1", {"MIRGE_WALKS": "2,2,1,1,1"}), ("3,1,2,1", {"MIRGE_WALKS": "3,1,2,1"})]
if os.environ.get("AB_WALKS"):
    settings = [("default", {})] + [(w, {"MIRGE_WALKS": w}) for w in os.environ["AB_WALKS"].split(";") if w != "x"]
if var:
    settings.append(("variant.so", {"MIRGE_NATIVE_SO": var}))
for item in filter(None, os.environ.get("AB_SOS", "").split(";")):  # AB_SOS="name=path.so;name2=path2.so[@ENV=VALUE...]"
    name, _, rest = item.partition("=")
    path, *envs = rest.split("@")
    settings.append((name, dict([("MIRGE_NATIVE_SO", os.path.abspath(path))] + [tuple(e.split("=", 1)) for e in envs])))
res = {n: [] for n, _ in settings}
for r in range(rounds):
    for name, env_add in settings:
        env = dict(os.environ, **env_add)
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "30", "--warmup", "3", "--cpu-baseline", "0", "--pmc", "0", "--two-in-flight", "0",
                              "--cli-path", "0", "--min-seconds", "1"], env=env, capture_output=True, text=True)
        try:
            d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        except Exception:
            print(name, "FAILED", out.stderr[-800:], flush=True)
            continue
        k = d["kernels"].get("k_cascade_bulk.w1", {}).get("avg_ms")
        res[name].append((d["ms_per_step"], k))
        print(name, d["ms_per_step"], "bulk", k, "timed", d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], flush=True)
for n, v in res.items():
    if v:
        print(f"{n:12s} step min {min(x[0] for x in v):.4f} median {sorted(x[0] for x in v)[len(v) // 2]:.4f}   bulk {[x[1] for x in v]}")
