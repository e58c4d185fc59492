"""A/B of two builds of libmirge_native.so in fresh processes, interleaved.
usage: python tools/ab_so.py <variant.so> [rounds]   (compares against the in-tree library)"""
import json, os, subprocess, sys
var = os.path.abspath(sys.argv[1]); rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {"default": [], "variant": []}
for r in range(rounds):
    for name in ("default", "variant"):
        env = dict(os.environ)
        if name == "variant": env["MIRGE_NATIVE_SO"] = var
        else: env.pop("MIRGE_NATIVE_SO", None)
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "30", "--warmup", "3", "--cpu-baseline", "0", "--pmc", "0", "--two-in-flight", "0"],
                             env=env, capture_output=True, text=True)
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        res[name].append(d["ms_per_step"])
        print(name, d["ms_per_step"], {k[6:]: round(v["avg_ms"], 3) for k, v in d["kernels"].items() if k.endswith(".w1") and "pass" in k}, flush=True)
print({k: (min(v), sorted(v)[len(v) // 2]) for k, v in res.items()})
