"""tests/golden/gangfit_golden_v4.json (tests/golden/make_golden_v4.py): the snapshot functions, the executor reschedule path
and the average packing efficiencies — the rows v1-v3 leave to restatements.  The restatements must keep reproducing the file
(CPU), the HIP path must reproduce it through the C ABI (GPU), and integration/go/golden_snapshot_test.go runs the REFERENCE's
own functions on the same file (and can regenerate it with -update) on a machine with a Go toolchain."""
import json
import os

import numpy as np
import pytest

from oracle import binding as ob
from oracle import pysnapshot as ps

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "gangfit_golden_v4.json")) as f:
    GOLDEN = json.load(f)
CASES = GOLDEN["cases"]
NO_NODE = 0xFFFFFFFF


def _cluster(c):
    g = lambda k, dt: np.asarray(c[k], dtype=dt)  # noqa: E731
    return dict(alloc=g("alloc", np.int64), overhead=g("overhead", np.int64), res_node=g("res_node", np.uint32),
                res_req=g("res_req", np.int64).reshape(-1, 3), node_flags=g("node_flags", np.uint32), name_rank=g("name_rank", np.uint32),
                zone=g("zone", np.uint32), n_zones=int(c["n_zones"]))


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_restatements_reproduce_v4(ci):
    c = CASES[ci]
    cl = _cluster(c)
    avail, sched, D, X = ps.build(**cl)
    s = c["snapshot"]
    assert avail.tolist() == s["avail"] and sched.tolist() == s["sched"] and D.tolist() == s["D"] and X.tolist() == s["X"]
    e = c["executor"]
    hosts = np.zeros((len(e["exe"]), len(avail)), dtype=np.uint8)
    for i, h in enumerate(e["hosts"]):
        hosts[i, h] = 1
    assert [ob.executor_fit(avail, x, X, reserved=e["first_fit_extra_reserved"]) for x in e["exe"]] == e["first_fit"]
    assert [ob.executor_fit(avail, x, X, reserved=cl["overhead"], minimal_fragmentation=True, hosts=hosts[i])
            for i, x in enumerate(e["exe"])] == e["minimal_fragmentation"]
    if "efficiency" in c:
        f = c["efficiency"]
        out = ob.fit_independent(0, avail, ob.make_apps(f["drv"], f["exe"], f["k"]), D, X, sched=sched, zone=cl["zone"])
        assert out.results["has_capacity"].tolist() == f["has_capacity"]
        assert [[f"{int(b):016x}" for b in row] for row in out.avg_eff.view(np.uint64)] == f["avg_bits"]


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(len(CASES)))
def test_hip_path_reproduces_v4(gf_ctx, ci):
    import gangfit

    c = CASES[ci]
    cl = _cluster(c)
    D, X = gf_ctx.build_snapshot(**cl)
    s = c["snapshot"]
    avail, sched = gf_ctx.snapshot()
    assert avail.tolist() == s["avail"] and sched.tolist() == s["sched"]
    assert D.tolist() == s["D"] and X.tolist() == s["X"]
    e = c["executor"]
    exe = np.asarray(e["exe"], dtype=np.int64)
    n = len(avail)
    assert gf_ctx.executor_fit(exe, reserved=np.asarray(e["first_fit_extra_reserved"], dtype=np.int64)).tolist() == e["first_fit"]
    hosts = np.zeros((len(exe), n), dtype=bool)
    for i, h in enumerate(e["hosts"]):
        hosts[i, h] = True
    got = gf_ctx.executor_fit(exe, reserved=cl["overhead"], minimal_fragmentation=True, hosts=hosts)
    assert got.tolist() == e["minimal_fragmentation"]
    if "efficiency" in c:
        f = c["efficiency"]
        apps = gangfit.make_apps(f["drv"], f["exe"], f["k"])
        out = gf_ctx.fit_batch(gangfit.GF_MODE_INDEPENDENT, 0, apps)
        assert out.results["has_capacity"].tolist() == f["has_capacity"]
        assert out.results["driver_node"].tolist() == f["driver_node"]
        avg = gf_ctx.avg_packing_efficiency(0, apps, out)
        assert [[f"{int(b):016x}" for b in row] for row in np.ascontiguousarray(avg).view(np.uint64)] == f["avg_bits"]
