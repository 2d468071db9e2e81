"""Randomised parity of the node-range-sharded batch (gangfit_shard.inc) against the oracle for a number of seconds: clusters whose
gpu nodes are a clumped minority (the compact gpu view exists and the shards' parts of it are ragged) and clusters with gpus
everywhere (no view), 1 .. 8 shards of one MI355X as a thread group, both packers.
    python tools/stress_sharded.py [seconds] [first seed]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd"), os.path.join(REPO, "tests")]
import gangfit
from oracle import binding as ob
from test_gpu_parity import _assert_same, _random_problem
from test_gpu_sharded import _run

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 31001
t0, cases, fit, nofit = time.time(), 0, 0, 0
while time.time() - t0 < seconds:
    rng = np.random.default_rng(seed)
    n = int(rng.choice([5, 64, 65, 130, 700, 2000, 5000]))
    world = int(rng.integers(1, 9))
    algo = int(rng.integers(0, 2))
    layout = "merged" if rng.random() < 0.7 else "identical"
    avail, D, X, drv, exe, k = _random_problem(rng, n, 150, bool(rng.random() < 0.5), layout)
    if rng.random() < 0.75:  # gpus on a clumped minority of the priority order
        avail[:, 2] = 0
        pos = np.arange(len(X))
        stride = max(1, len(X) // int(rng.integers(2, 12)))
        clump = ((pos // stride) % 3 == int(rng.integers(0, 3))) & (rng.random(len(X)) < rng.uniform(0.05, 0.6))
        nodes = X[clump]
        nodes = nodes[nodes < n]
        avail[nodes, 2] = rng.integers(1, 9, size=len(nodes))
        exe[:, 2] = np.where(rng.random(len(exe)) < 0.7, rng.integers(1, 4, size=len(exe)), 0)
        drv[:, 2] = np.where(rng.random(len(drv)) < 0.3, 1, 0)
    small = rng.random(len(k)) < 0.75
    k = np.where(small, np.minimum(k, rng.integers(0, 40, size=len(k))), k).astype(np.int32)
    apps = gangfit.make_apps(drv, exe, k)
    ref = ob.fit_independent(algo, avail, ob.make_apps(drv, exe, k), D, X, closed_form=True)
    try:
        for out in _run(world, algo, avail, D, X, apps):
            _assert_same(out, ref, apps)
    except Exception as e:
        print(f"MISMATCH seed {seed} n {n} world {world} algo {algo} layout {layout}: {type(e).__name__}: {e}", flush=True)
        sys.exit(1)
    g = (exe[:, 2] > 0) & (k > 0)
    fit += int((ref.results["has_capacity"][g] != 0).sum())
    nofit += int((ref.results["has_capacity"][g] == 0).sum())
    cases += 1
    seed += 1
print(f"sharded stress ok: {cases} cases ({fit} gangs of gpu executors that fit, {nofit} that do not) over seeds "
      f"{seed - cases} .. {seed - 1} in {time.time() - t0:.0f} s")
