"""Interleaved A/B timing of differently-built library variants on ONE device
(cdna_hip_programming.md 5.4 rule 24): for each round, for each variant, run
bench.py in a child process with GP_PREDICT_LIB pointing at that variant.

    python tools/ab_bench.py --rounds 3 [--args "--workload c2"] name=path.so name2=path2.so
"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--args", default="")
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
vs = [v.split("=", 1) for v in a.variants]
res = {n: [] for n, _ in vs}
for r in range(a.rounds):
    for n, path in vs:
        env = dict(os.environ, GP_PREDICT_LIB=os.path.abspath(path))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline",
                              "--no-e2e", "--steps", "20", "--warmup", "3"] + a.args.split(),
                             env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(n, "FAILED", out.stderr[-500:])
            continue
        res[n].append(json.loads(line[-1])["roofline"]["kernel_ms"])
        print("round %d %-24s kernel_ms %.4f" % (r, n, res[n][-1]), flush=True)
for n, v in res.items():
    if v:
        print("%-24s min %.4f median %.4f (n=%d)" % (n, min(v), sorted(v)[len(v) // 2], len(v)))
