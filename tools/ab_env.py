"""A/B of one environment switch in fresh processes, interleaved (tools only; not part of the product).
usage: python tools/ab_env.py VAR A B [rounds]"""
import json, os, subprocess, sys
var, a, b = sys.argv[1:4]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {a: [], b: []}
for r in range(rounds):
    for v in (a, b):
        env = dict(os.environ, **{var: v})
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "3", "--cpu-baseline", "0", "--pmc", "0", "--cli-path", "0"],
                             env=env, capture_output=True, text=True)
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        res[v].append(d["ms_per_step"])
        print(var, v, d["ms_per_step"], d["timing"]["step_ms_rank0"]["median"],
              {k[6:]: round(x["avg_ms"], 3) for k, x in d["kernels"].items() if k.endswith(".w1") and ("pass" in k or "cascade" in k)}, flush=True)
print({k: (min(v), sorted(v)[len(v) // 2]) for k, v in res.items()})
