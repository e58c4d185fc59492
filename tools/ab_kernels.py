"""Per-kernel A/B of two builds: python tools/ab_kernels.py <variant.so> [rounds] -> avg_ms of every kernel, default vs variant."""
import json, os, subprocess, sys
var = os.path.abspath(sys.argv[1]); rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = {"default": {}, "variant": {}}
step = {"default": [], "variant": []}
for r in range(rounds):
    for name in ("default", "variant"):
        env = dict(os.environ)
        if name == "variant": env["MIRGE_NATIVE_SO"] = var
        else: env.pop("MIRGE_NATIVE_SO", None)
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "30", "--warmup", "3", "--cpu-baseline", "0", "--pmc", "0", "--two-in-flight", "0", "--cli-path", "0"],
                             env=env, capture_output=True, text=True)
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        step[name].append(d["ms_per_step"])
        for k, v in d["kernels"].items():
            acc[name].setdefault(k, []).append(v["avg_ms"] * v["launches"] / 2)
print("ms_per_step", {k: [round(x, 4) for x in v] for k, v in step.items()})
tot = {"default": 0.0, "variant": 0.0}
for k in sorted(acc["default"]):
    a = sum(acc["default"][k]) / len(acc["default"][k]); b = sum(acc["variant"].get(k, [0])) / max(len(acc["variant"].get(k, [0])), 1)
    tot["default"] += a; tot["variant"] += b
    if max(a, b) > 0.004: print(f"{k:28s} {a:8.4f} {b:8.4f} {b - a:+.4f}")
print("sum", {k: round(v, 4) for k, v in tot.items()})
