"""GPU parity of the node-range sharded batch (csrc/gangfit_shard.inc through include/gangfit.h's gf_shard_* entry
points, driven by gangfit/sharded.py): several shards of ONE MI355X (a thread group, one gf_ctx per shard) and a
world_size-2 gloo process group sharing cuda:0, against the oracle's unsharded answer.  `python -m pytest tests -m gpu`."""
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
from test_gpu_parity import _assert_same, _random_problem

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, algo, avail, D, X, apps):
    import torch

    # the HIP runtime and torch's device count are initialised HERE, by one thread: `world` threads that meet in the first
    # torch.cuda.is_available() of a process — next to the first gf_init of their siblings — have seen "no device" on the MI355X box
    # (tools/stress_sharded.py when its first seed drew eight shards: gpurun_out/r6av, r6aw), and torch caches that answer
    torch.cuda.init()
    group = sharded.ThreadGroup(world)
    outs, errs = [None] * world, []

    def work(r):
        try:
            with gangfit.Context(0) as ctx:
                ctx.set_snapshot(avail)
                ctx.set_orders(D, X)
                eng = sharded.HipShardEngine(ctx, r, world, "cuda:0")
                outs[r] = sharded.sharded_fit(eng, group.comm(r), algo, apps)
        except Exception as e:
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
@pytest.mark.parametrize("n", [5, 64, 130, 1000])
def test_shards_of_one_gpu_match_oracle(algo, world, n):
    rng = np.random.default_rng(1234 + 31 * world + algo + n)
    for layout in ("merged", "identical"):
        for tight_cluster in (True, False):
            avail, D, X, drv, exe, k = _random_problem(rng, n, 150, tight_cluster, layout)
            apps = gangfit.make_apps(drv, exe, k)
            ref = ob.fit_independent(algo, avail, ob.make_apps(drv, exe, k), D, X, closed_form=True)
            for out in _run(world, algo, avail, D, X, apps):
                _assert_same(out, ref, apps)


@pytest.mark.parametrize("algo", [0, 1])
@pytest.mark.parametrize("world", [1, 2, 3, 5, 8])
def test_gpu_gangs_take_the_ranges_part_of_the_compact_view(algo, world):
    """Clusters whose gpu nodes are a minority (the sparse gpu view exists: gangfit_api_snapshot.cpp) and sit in clumps, so that
    the shards' parts of the compact table are ragged — empty for some ranges, cut in the middle of a 64-lane chunk for
    others —, gangs of gpu executors that fit, that do not, and whose driver lands on a gpu node (gangfit_shard.inc)."""
    rng = np.random.default_rng(777 + 13 * world + algo)
    seen = [0, 0]  # gangs of gpu executors that fit / that do not
    for n in (70, 700, 3000):
        for tight_cluster in (True, False):
            avail, D, X, drv, exe, k = _random_problem(rng, n, 200, tight_cluster, "merged")
            # gpus on ~12 % of the nodes, in clumps of the PRIORITY order (X), none anywhere else
            avail[:, 2] = 0
            pos = np.arange(len(X))
            clump = ((pos // max(1, len(X) // 9)) % 3 == 1) & (rng.random(len(X)) < 0.4)
            nodes = X[clump]
            nodes = nodes[nodes < n]
            avail[nodes, 2] = rng.integers(1, 9, size=len(nodes))
            exe[:, 2] = np.where(rng.random(len(exe)) < 0.7, rng.integers(1, 4, size=len(exe)), 0)
            drv[:, 2] = np.where(rng.random(len(drv)) < 0.3, 1, 0)
            small = rng.random(len(k)) < 0.75  # the others keep gang sizes of up to three times the cluster
            k = np.where(small, np.minimum(k, rng.integers(0, 40, size=len(k))), k).astype(np.int32)
            apps = gangfit.make_apps(drv, exe, k)
            ref = ob.fit_independent(algo, avail, ob.make_apps(drv, exe, k), D, X, closed_form=True)
            gpu_gang = (exe[:, 2] > 0) & (k > 0)
            seen[0] += int((ref.results["has_capacity"][gpu_gang] != 0).sum())
            seen[1] += int((ref.results["has_capacity"][gpu_gang] == 0).sum())
            for out in _run(world, algo, avail, D, X, apps):
                _assert_same(out, ref, apps)
    assert seen[0] > 20 and seen[1] > 20


def test_general_layout_is_refused():
    with gangfit.Context(0) as ctx:
        ctx.set_snapshot([[5, 5, 0], [5, 5, 0]])
        ctx.set_orders([0, 1], [1, 0])  # the two orders disagree: no merged order exists
        eng = sharded.HipShardEngine(ctx, 0, 2, "cuda:0")
        with pytest.raises(gangfit.GangfitError) as e:
            sharded.sharded_fit(eng, sharded.SingleComm(), 0, gangfit.make_apps([[1, 1, 0]], [[1, 1, 0]], [1]))
        assert e.value.code == gangfit._native.GF_ERR_UNSUPPORTED


@pytest.mark.parametrize("congested", [False, True])
def test_headline_size_eight_shards(congested):
    w = wl.headline(10000, 1000, congested=congested)
    s = w.snapshot
    apps = gangfit.make_apps(w.drv, w.exe, w.k)
    for algo in (0, 1):
        ref = ob.fit_independent(algo, s.avail, ob.make_apps(w.drv, w.exe, w.k), s.driver_order, s.exec_order,
                                 closed_form=True)
        outs = _run(8, algo, s.avail, s.driver_order, s.exec_order, apps)
        _assert_same(outs[0], ref, apps)
        _assert_same(outs[7], ref, apps)


_WORKER = r"""
import os, sys
import numpy as np
sys.path[:0] = [{repo!r}, os.path.join({repo!r}, "k8s-spark-scheduler_amd"), os.path.join({repo!r}, "tests")]
import torch, torch.distributed as dist
import gangfit
from gangfit import sharded, workloads as wl
from oracle import binding as ob
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
comm = sharded.TorchComm()
w = wl.config(2, n_nodes=3000, n_apps=400)
s = w.snapshot
ok = True
with gangfit.Context(0) as ctx:
    ctx.set_snapshot(s.avail)
    ctx.set_orders(s.driver_order, s.exec_order)
    eng = sharded.HipShardEngine(ctx, comm.rank, comm.world, "cuda:0")
    apps = gangfit.make_apps(w.drv, w.exe, w.k)
    for algo in (0, 1):
        out = sharded.sharded_fit(eng, comm, algo, apps)
        ref = ob.fit_independent(algo, s.avail, ob.make_apps(w.drv, w.exe, w.k), s.driver_order, s.exec_order, closed_form=True)
        ok = ok and np.array_equal(out.results, ref.results)
        for a in np.nonzero(ref.results["has_capacity"])[0]:
            ok = ok and np.array_equal(out.placement(int(a))[2], ref.placement(int(a))[2])
dist.barrier()
dist.destroy_process_group()
print("SHARDED_OK" if ok else "SHARDED_MISMATCH")
"""


def test_two_processes_gloo_sharing_the_gpu():
    port = 31500 + (os.getpid() % 2000)
    code = _WORKER.format(repo=REPO, port=port)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "SHARDED_OK" in o, o[-2000:]
