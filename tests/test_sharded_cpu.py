"""N > 1 control flow of the node-range sharded batch (gangfit/sharded.py) on CPU: the numpy stand-in engine from
tests/shard_ref_engine.py under (a) an in-process thread group and (b) a real world_size-2 gloo process group, checked
against the oracle's unsharded answer."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import gangfit
from gangfit import sharded
from gangfit import workloads as wl
from oracle import binding as ob
from shard_ref_engine import RefShardEngine

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(seed, n, a, congested):
    rng = np.random.default_rng(seed)
    hi = 12 if congested else 400
    avail = rng.integers(-2, hi, size=(n, 3)).astype(np.int64)
    order = rng.permutation(n)
    xflag = rng.random(n) < 0.85
    dflag = rng.random(n) < 0.7
    dflag[order.tolist().index(order[-1])] = True
    D = order[dflag].astype(np.uint32)
    X = order[xflag].astype(np.uint32)
    drv = rng.integers(0, 9, size=(a, 3)).astype(np.int64)
    exe = rng.integers(0, 6, size=(a, 3)).astype(np.int64)
    exe[rng.random(a) < 0.5, 2] = 0
    k = rng.integers(0, 3 * n if congested else n, size=a).astype(np.int32)
    k[~exe.any(axis=1)] = np.minimum(k[~exe.any(axis=1)], 100)
    return avail, order, xflag, dflag, D, X, drv, exe, k


def _run_threads(world, algo, avail, order, xflag, dflag, apps):
    group = sharded.ThreadGroup(world)
    outs = [None] * world
    errs = []

    def work(r):
        try:
            eng = RefShardEngine(avail, order, xflag, dflag, r, world)
            outs[r] = sharded.sharded_fit(eng, group.comm(r), algo, apps)
        except Exception as e:  # pragma: no cover
            errs.append(e)
            group._barrier.abort()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    if errs:
        raise errs[0]
    return outs


@pytest.mark.parametrize("algo", [0, 1])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_thread_group_matches_oracle(algo, world):
    for congested in (False, True):
        avail, order, xflag, dflag, D, X, drv, exe, k = _problem(17 * world + algo, 300, 60, congested)
        apps = gangfit.make_apps(drv, exe, k)
        ref = ob.fit_independent(algo, avail, ob.make_apps(drv, exe, k), D, X, closed_form=True)
        outs = _run_threads(world, algo, avail, order, xflag, dflag, apps)
        for o in outs:  # every rank ends with the complete answer
            assert np.array_equal(o.results, ref.results)
            for a in np.nonzero(ref.results["has_capacity"])[0]:
                assert np.array_equal(o.placement(int(a))[2], ref.placement(int(a))[2]), (a, world, congested)
        if congested:  # both outcomes, and distribute-evenly beyond pass 1, are exercised
            assert ref.results["has_capacity"].any() and not ref.results["has_capacity"].all()


_WORKER = r"""
import os, sys
import numpy as np
sys.path[:0] = [{repo!r}, os.path.join({repo!r}, "k8s-spark-scheduler_amd"), os.path.join({repo!r}, "tests")]
import torch.distributed as dist
import gangfit
from gangfit import sharded
from oracle import binding as ob
from shard_ref_engine import RefShardEngine
from test_sharded_cpu import _problem
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
comm = sharded.TorchComm()
ok = True
for algo in (0, 1):
    for congested in (False, True):
        avail, order, xflag, dflag, D, X, drv, exe, k = _problem(99 + algo, 200, 40, congested)
        apps = gangfit.make_apps(drv, exe, k)
        out = sharded.sharded_fit(RefShardEngine(avail, order, xflag, dflag, comm.rank, comm.world), comm, algo, apps)
        ref = ob.fit_independent(algo, avail, ob.make_apps(drv, exe, k), D, X, closed_form=True)
        ok = ok and np.array_equal(out.results, ref.results)
        for a in np.nonzero(ref.results["has_capacity"])[0]:
            ok = ok and np.array_equal(out.placement(int(a))[2], ref.placement(int(a))[2])
dist.barrier()
dist.destroy_process_group()
print("SHARDED_OK" if ok else "SHARDED_MISMATCH")
"""


def test_world_size_2_gloo():
    port = 29500 + (os.getpid() % 2000)
    code = _WORKER.format(repo=REPO, port=port)
    env = dict(os.environ, GANGFIT_NO_TORCH="0")
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              env=env, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "SHARDED_OK" in o, o[-2000:]
