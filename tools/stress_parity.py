"""Randomised parity stress (run on the MI355X box): seeds x packers x layouts x contexts against the oracle (tests/stress_lib.py).
    python tools/stress_parity.py [seconds] [first seed]
Prints the first mismatch with its seed and exits 1, or a summary and exits 0.  A progress line every 25 seeds (flushed): a run
that is cut off from outside still says how far it got.  The budget is checked between seeds — a seed whose oracle is slow (the
literal driver retry loop on a large cluster with gangs that do not fit) can overrun it by a minute or two."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd"), os.path.join(REPO, "tests")]
import stress_lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctxs = stress_lib.make_contexts()
t_end = time.time() + budget
n_cases, first = 0, seed
while time.time() < t_end:
    c, bad = stress_lib.one_seed(ctxs, seed)
    n_cases += c
    if bad:
        print(bad)
        sys.exit(1)
    seed += 1
    if (seed - first) % 25 == 0:
        print(f"... {n_cases} cases green, next seed {seed}, {t_end - time.time():.0f} s left", flush=True)
for c in ctxs.values():
    c.close()
print(f"stress ok: {n_cases} (context, packer) cases over seeds {first} .. {seed - 1} in {budget:.0f} s")
