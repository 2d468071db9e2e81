"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same inputs — bit-exact
(integer/index work).  Run on the MI355X box: `python -m pytest tests -m gpu`."""
import numpy as np
import pytest

import gangfit
import kats
from gangfit import workloads as wl
from oracle import binding as ob

pytestmark = pytest.mark.gpu

TIGHT, EVEN = gangfit.GF_ALGO_TIGHTLY_PACK, gangfit.GF_ALGO_DISTRIBUTE_EVENLY
IND, FIFO = gangfit.GF_MODE_INDEPENDENT, gangfit.GF_MODE_FIFO_CHAIN


def _assert_same(gpu: gangfit.BatchOut, ref: ob.BatchOut, apps):
    assert np.array_equal(gpu.results["has_capacity"], ref.results["has_capacity"])
    assert np.array_equal(gpu.results["evaluated"], ref.results["evaluated"])
    assert np.array_equal(gpu.results["driver_node"], ref.results["driver_node"])
    assert np.array_equal(gpu.results["exec_len"], ref.results["exec_len"])
    # placements only where feasible (infeasible apps leave their slice unspecified on both sides)
    feas = np.nonzero(ref.results["has_capacity"])[0]
    for a in feas:
        g, r = gpu.placement(int(a))[2], ref.placement(int(a))[2]
        assert np.array_equal(g, r), f"app {a}: k={apps['k'][a]} gpu={g[:16]} ref={r[:16]}"


def _gpu_apps(drv, exe, k, flags=None):
    return gangfit.make_apps(drv, exe, k, flags)


def test_device_is_gfx950(gf_ctx):
    info = gf_ctx.device_info()
    assert info["arch"].startswith("gfx950") and info["wavefront_size"] == 64
    assert info["compute_units"] >= 200


def test_hbm_stream_copy_probe(gf_ctx):
    """The roofline's measured companion of the 8 TB/s spec figure: a plain copy must move terabytes per second."""
    rd, cp = gf_ctx.hbm_probe(1 << 30, 4)
    assert rd > 1000.0 and cp > 1000.0
    assert 0.5 < gf_ctx.launch_floor(0, 100) < 100.0


def test_wave_primitives_selftest(gf_ctx):
    # DPP prefix scan and the f64-reciprocal exact division vs plain serial code / 64-bit divide, on device
    assert gf_ctx.selftest(seed=1, n_cases=512) == 0
    assert gf_ctx.selftest(seed=0xDEADBEEF, n_cases=512) == 0


@pytest.mark.parametrize("case", kats.ALL, ids=[c["name"] for c in kats.ALL])
def test_known_answers(gf_ctx, case):
    gf_ctx.set_snapshot(case["avail"])
    gf_ctx.set_orders(case["D"], case["X"])
    ok, driver, execs = gf_ctx.spark_binpack(case["algo"], case["drv"], case["exe"], case["k"])
    assert ok == case["feasible"]
    if ok:
        assert driver == case["driver"] and execs.tolist() == case["execs"]
    else:
        assert driver == gangfit.GF_NO_NODE and len(execs) == 0


def test_fifo_quirk_k7(gf_ctx):
    c = kats.FIFO_K7
    gf_ctx.set_snapshot(c["avail"])
    gf_ctx.set_orders(c["D"], c["X"])
    apps = _gpu_apps([a["drv"] for a in c["apps"]], [a["exe"] for a in c["apps"]], [a["k"] for a in c["apps"]])
    out = gf_ctx.fit_batch(FIFO, TIGHT, apps)
    ok, driver, execs = out.placement(0)
    assert ok and driver == c["first"]["driver"] and execs.tolist() == c["first"]["execs"]
    assert gf_ctx.residual().tolist() == c["residual"]
    assert out.failed_at == -1 and not out.placement(1)[0]


def _random_problem(rng, n, a, tight_cluster, layout="general"):
    """layout: "general" = driver and executor orders are independent permutations (the two orders disagree);
    "merged" = both are subsequences of one priority order, as NodeSorter.PotentialNodes produces them (driver-only and
    executor-only nodes, unknown names and a repeated driver candidate included); "identical" = D == X."""
    hi = 40 if tight_cluster else 4000
    avail = rng.integers(-3, hi, size=(n, 3)).astype(np.int64)
    avail[:, 2] = rng.integers(-1, 9, size=n)
    unknown = np.array([n + 5, n + 1000], dtype=np.int64)
    if layout == "general":
        X = np.concatenate([rng.permutation(n)[: int(rng.integers(max(1, n // 2), n + 1))], unknown[:1]])
        X = rng.permutation(X).astype(np.uint32)
        D = np.concatenate([rng.permutation(n)[: int(rng.integers(1, n + 1))], unknown])
        D = rng.permutation(D).astype(np.uint32)
    else:
        base = rng.permutation(n)
        if layout == "identical":
            X = base.astype(np.uint32)
            D = X.copy()
        else:
            X = base[rng.random(n) < 0.8]
            D = base[rng.random(n) < 0.7]
            if len(X) == 0:
                X = base[:1]
            if len(D) == 0:
                D = base[-1:]
            # unknown names anywhere, one repeated driver candidate at the end
            X = np.insert(X, int(rng.integers(0, len(X) + 1)), unknown[0]).astype(np.uint32)
            D = np.insert(D, int(rng.integers(0, len(D) + 1)), unknown[1])
            D = np.append(D, D[0]).astype(np.uint32)
    drv = rng.integers(0, 9, size=(a, 3)).astype(np.int64)
    exe = rng.integers(0, 7, size=(a, 3)).astype(np.int64)
    exe[rng.random(a) < 0.5, 2] = 0
    k = rng.integers(0, 3 * n, size=a).astype(np.int32)
    if tight_cluster:
        k = np.minimum(k, rng.integers(0, 200, size=a)).astype(np.int32)
    zero_exe = ~exe.any(axis=1)
    k[zero_exe] = np.minimum(k[zero_exe], 300)
    return avail, D, X, drv, exe, k


@pytest.mark.parametrize("layout", ["general", "merged", "identical"])
@pytest.mark.parametrize("algo", [TIGHT, EVEN])
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 128, 129, 500, 1000])
def test_independent_batch_random(gf_ctx, algo, n, layout):
    rng = np.random.default_rng(1000 * algo + n + 7 * len(layout))
    for tight_cluster in (True, False):
        a = 257
        avail, D, X, drv, exe, k = _random_problem(rng, n, a, tight_cluster, layout)
        gf_ctx.set_snapshot(avail)
        gf_ctx.set_orders(D, X)
        apps = _gpu_apps(drv, exe, k)
        gpu = gf_ctx.fit_batch(IND, algo, apps)
        ref = ob.fit_independent(algo, avail, ob.make_apps(drv, exe, k), D, X)
        _assert_same(gpu, ref, apps)
        if n >= 63 and layout != "merged":  # the generator must exercise both outcomes
            assert ref.results["has_capacity"].any()


@pytest.mark.parametrize("layout", ["general", "merged", "identical"])
@pytest.mark.parametrize("algo", [TIGHT, EVEN])
@pytest.mark.parametrize("n", [3, 64, 200, 1000, 1500, 5000])
def test_fifo_chain_random(gf_ctx, algo, n, layout):
    rng = np.random.default_rng(77 * (algo + 1) + n + 7 * len(layout))
    for rep in range(3):
        a = 120
        avail, D, X, drv, exe, k = _random_problem(rng, n, a, tight_cluster=(rep == 2), layout=layout)
        exe = np.maximum(exe, 1)  # keep chains long: every placement consumes something
        k = np.minimum(k, 40).astype(np.int32)
        flags = (rng.random(a) < (0.9 if rep else 1.0)).astype(np.uint32)  # rep 0: nothing aborts the chain
        gf_ctx.set_snapshot(avail)
        gf_ctx.set_orders(D, X)
        apps = _gpu_apps(drv, exe, k, flags)
        gpu = gf_ctx.fit_batch(FIFO, algo, apps)
        ref = ob.fit_fifo_chain(algo, avail, ob.make_apps(drv, exe, k, flags), D, X)
        assert gpu.failed_at == ref.failed_at
        _assert_same(gpu, ref, apps)
        assert np.array_equal(gf_ctx.residual(), ref.avail_after)


_CHAIN_VARIANTS = [{}, {"lds_budget": 24000}, {"fifo_generic": 1}, {"chain_cache": 0}]
_CHAIN_IDS = ["solo", "solo-global-tail", "wide-v2", "solo-no-chain-cache"]


def _ctx_with_env(options):
    return gangfit.Context(0, options=options)


@pytest.mark.parametrize("env", _CHAIN_VARIANTS, ids=_CHAIN_IDS)
@pytest.mark.parametrize("algo", [TIGHT, EVEN])
def test_fifo_chain_kernel_variants(algo, env):
    """Every chain kernel of the plain packers on the same problems (merged layout): the one-controlling-wavefront chain
    (gangfit_fifo_solo.inc; whole table in LDS, and with a global-memory tail) and the wide kernel.  Cases: few request shapes (exact chunk index), more than 64 distinct shapes (chunk-maxima path),
    a depleting cluster where most of the queue cannot fit (capacity bound learnt per shape, driver fallback, several
    distribute-evenly passes), a request without a scaled form (wide fallback)."""
    ctx = _ctx_with_env(env)
    rng = np.random.default_rng(4242 + algo)
    try:
        for rep in range(6):
            n, a = ((3000, 150), (900, 200), (5000, 120), (700, 260), (2000, 150), (1500, 100))[rep]
            avail, D, X, drv, exe, k = _random_problem(rng, n, a, tight_cluster=rep in (1, 3), layout="merged")
            exe = np.maximum(exe, 1)
            k = np.minimum(k, 60 if rep != 2 else 700).astype(np.int32)
            if rep in (0, 1, 2):  # a handful of templates: few distinct shapes, long runs of the same shape
                t = rng.integers(0, 5, size=a)
                drv, exe = drv[t], exe[t]
            if rep == 3:  # the cluster runs dry: most apps are skipped, the same shapes keep coming back
                t = rng.integers(0, 3, size=a)
                drv, exe = drv[t], exe[t]
                k = rng.integers(20, 200, size=a).astype(np.int32)
            if rep == 4:  # more than 64 distinct request vectors
                drv = rng.integers(0, 40, size=(a, 3)).astype(np.int64)
                exe = rng.integers(1, 40, size=(a, 3)).astype(np.int64)
                avail = avail * 8
            flags = (rng.random(a) < (0.97 if rep in (0, 5) else 1.0)).astype(np.uint32)  # reps 1-4: nothing aborts the chain
            if rep == 5:
                avail = avail * 6
                drv[7, 1] = 5  # not a multiple of the table's unit
            ctx.set_snapshot(avail)
            ctx.set_orders(D, X)
            apps = _gpu_apps(drv, exe, k, flags)
            gpu = ctx.fit_batch(FIFO, algo, apps)
            ref = ob.fit_fifo_chain(algo, avail, ob.make_apps(drv, exe, k, flags), D, X)
            assert gpu.failed_at == ref.failed_at, f"rep {rep}"
            _assert_same(gpu, ref, apps)
            assert np.array_equal(ctx.residual(), ref.avail_after), f"rep {rep}"
            if rep == 3:
                assert 0.02 < ref.results["has_capacity"].mean() < 0.9
    finally:
        ctx.close()


@pytest.mark.parametrize("env", [{}, {"chain_cache": 0}], ids=["solo", "solo-no-chain-cache"])
@pytest.mark.parametrize("algo", [TIGHT, EVEN])
def test_fifo_chain_beyond_the_lds_front(algo, env):
    """24 000 nodes: the table does not fit the LDS front (global tail) and a shape's index row has more than four words
    (the words beyond the first four are read on demand) — against the closed-form oracle, nominal and congested."""
    ctx = _ctx_with_env(env)
    try:
        for congested in (False, True):
            w = wl.headline(24000, 400, congested=congested)
            s = w.snapshot
            ctx.set_snapshot(s.avail, s.sched)
            ctx.set_orders(s.driver_order, s.exec_order)
            apps = _gpu_apps(w.drv, w.exe, w.k, np.ones(len(w.k), dtype=np.uint32))
            oapps = ob.make_apps(w.drv, w.exe, w.k, np.ones(len(w.k), dtype=np.uint32))
            gpu = ctx.fit_batch(FIFO, algo, apps)
            ref = ob.fit_fifo_chain(algo, s.avail, oapps, s.driver_order, s.exec_order, closed_form=True)
            assert gpu.failed_at == ref.failed_at == -1
            _assert_same(gpu, ref, apps)
            assert np.array_equal(ctx.residual(), ref.avail_after)
    finally:
        ctx.close()


@pytest.mark.parametrize("n_apps", [1, 2, 31, 32, 33, 64, 65, 97])
def test_fifo_chain_staging_boundaries(gf_ctx, n_apps):
    """Chains whose length sits on the boundaries of the app-record staging (halves of 32 records)."""
    rng = np.random.default_rng(900 + n_apps)
    avail, D, X, drv, exe, k = _random_problem(rng, 700, n_apps, tight_cluster=False, layout="merged")
    exe = np.maximum(exe, 1)
    k = np.minimum(k, 30).astype(np.int32)
    k[:: 5] = 0  # driver-only applications in between
    flags = np.ones(n_apps, dtype=np.uint32)
    gf_ctx.set_snapshot(avail)
    gf_ctx.set_orders(D, X)
    for algo in (TIGHT, EVEN):
        apps = _gpu_apps(drv, exe, k, flags)
        gpu = gf_ctx.fit_batch(FIFO, algo, apps)
        ref = ob.fit_fifo_chain(algo, avail, ob.make_apps(drv, exe, k, flags), D, X)
        assert gpu.failed_at == ref.failed_at
        _assert_same(gpu, ref, apps)
        assert np.array_equal(gf_ctx.residual(), ref.avail_after)


@pytest.mark.parametrize("number", [1, 2])
@pytest.mark.parametrize("algo", [TIGHT, EVEN])
def test_baseline_configs_small(gf_ctx, number, algo):
    """BASELINE.json configs[0] and configs[1] (16 nodes; 1k nodes x 1k apps) bit-exact against the oracle."""
    w = wl.config(number)
    s = w.snapshot
    gf_ctx.set_snapshot(s.avail, s.sched)
    gf_ctx.set_orders(s.driver_order, s.exec_order)
    apps = _gpu_apps(w.drv, w.exe, w.k)
    gpu = gf_ctx.fit_batch(IND, algo, apps)
    ref = ob.fit_independent(algo, s.avail, ob.make_apps(w.drv, w.exe, w.k), s.driver_order, s.exec_order)
    _assert_same(gpu, ref, apps)
    gpu = gf_ctx.fit_batch(FIFO, algo, apps)
    ref = ob.fit_fifo_chain(algo, s.avail, ob.make_apps(w.drv, w.exe, w.k), s.driver_order, s.exec_order)
    assert gpu.failed_at == ref.failed_at
    _assert_same(gpu, ref, apps)
    assert np.array_equal(gf_ctx.residual(), ref.avail_after)


@pytest.mark.parametrize("congested", [False, True])
@pytest.mark.parametrize("algo", [TIGHT, EVEN])
def test_headline_size_against_closed_form_oracle(gf_ctx, algo, congested):
    """10k nodes x 1k pending apps (the size the metric is quoted on): independent batch and FIFO chain vs the
    closed-form oracle (the literal one is O(|D| N) per infeasible app), plus a literal spot check."""
    w = wl.headline(congested=congested)
    s = w.snapshot
    gf_ctx.set_snapshot(s.avail, s.sched)
    gf_ctx.set_orders(s.driver_order, s.exec_order)
    apps = _gpu_apps(w.drv, w.exe, w.k, w.flags)
    oapps = ob.make_apps(w.drv, w.exe, w.k, w.flags)
    gpu = gf_ctx.fit_batch(IND, algo, apps)
    ref = ob.fit_independent(algo, s.avail, oapps, s.driver_order, s.exec_order, closed_form=True)
    _assert_same(gpu, ref, apps)
    lit = ob.fit_independent(algo, s.avail, oapps[:16], s.driver_order, s.exec_order, closed_form=False)
    for a in range(16):
        assert gpu.placement(a)[0] == lit.placement(a)[0]
        if lit.placement(a)[0]:
            assert gpu.placement(a)[1] == lit.placement(a)[1]
            assert np.array_equal(gpu.placement(a)[2], lit.placement(a)[2])
    # FIFO: make every app skippable so the chain runs to the end even on the congested cluster
    apps["flags"] = 1
    oapps["flags"] = 1
    gpu = gf_ctx.fit_batch(FIFO, algo, apps)
    ref = ob.fit_fifo_chain(algo, s.avail, oapps, s.driver_order, s.exec_order, closed_form=True)
    assert gpu.failed_at == ref.failed_at == -1
    _assert_same(gpu, ref, apps)
    assert np.array_equal(gf_ctx.residual(), ref.avail_after)


@pytest.mark.parametrize("algo", [TIGHT, EVEN])
def test_headline_size_against_the_literal_oracle(gf_ctx, algo):
    """The size the metric is quoted on against the LITERAL restatement, end to end (all 1 000 applications, independent
    batch and FIFO chain with its residual table): on the nominal cluster every gang fits, so the literal loops — the ones
    bench.py times as the CPU baseline — finish in milliseconds.  (The congested cluster keeps the closed form above: an
    infeasible gang costs the literal retry loop O(|D| N).)"""
    w = wl.headline()
    s = w.snapshot
    gf_ctx.set_snapshot(s.avail, s.sched)
    gf_ctx.set_orders(s.driver_order, s.exec_order)
    apps = _gpu_apps(w.drv, w.exe, w.k, w.flags)
    oapps = ob.make_apps(w.drv, w.exe, w.k, w.flags)
    gpu = gf_ctx.fit_batch(IND, algo, apps)
    lit = ob.fit_independent(algo, s.avail, oapps, s.driver_order, s.exec_order, closed_form=False)
    _assert_same(gpu, lit, apps)
    assert lit.results["has_capacity"].all()
    apps["flags"] = 1
    oapps["flags"] = 1
    gpu = gf_ctx.fit_batch(FIFO, algo, apps)
    lit = ob.fit_fifo_chain(algo, s.avail, oapps, s.driver_order, s.exec_order, closed_form=False)
    assert gpu.failed_at == lit.failed_at == -1
    _assert_same(gpu, lit, apps)
    assert np.array_equal(gf_ctx.residual(), lit.avail_after)


def _check_properties(algo, w, out: gangfit.BatchOut):
    """Size-independent properties of a feasible placement (no oracle involved)."""
    s = w.snapshot
    pos_in_x = np.full(len(s.avail), -1, dtype=np.int64)
    pos_in_x[s.exec_order] = np.arange(len(s.exec_order))
    for a in range(len(w.k)):
        ok, driver, execs = out.placement(a)
        if not ok:
            continue
        assert len(execs) == w.k[a]
        assert np.all(w.drv[a] <= s.avail[driver])  # driver-fit check
        nodes, counts = np.unique(execs, return_counts=True)
        need = counts[:, None] * w.exe[a][None, :]
        need[nodes == driver] += w.drv[a]
        assert np.all(need <= s.avail[nodes]), "placement overcommits a node"
        p = pos_in_x[execs]
        assert np.all(p >= 0)
        if algo == TIGHT:
            assert np.all(np.diff(p) >= 0), "tightly-pack output must be node-major in priority order"
        else:
            # pass-major: within a pass positions strictly increase; a node's r-th copy appears in pass r
            passes = np.zeros(len(execs), dtype=np.int64)
            seen = {}
            for i, n in enumerate(execs):
                seen[n] = seen.get(n, 0) + 1
                passes[i] = seen[n]
            assert np.all(np.diff(passes) >= 0)
            same = np.diff(passes) == 0
            assert np.all(np.diff(p)[same] > 0)


@pytest.mark.parametrize("algo", [TIGHT, EVEN])
def test_config3_full_size_properties(gf_ctx, algo):
    """BASELINE.json configs[2] at full size (10k nodes x 10k apps): feasibility against the closed-form oracle and
    structural properties of every placement; a 64-app slice bit-exact against the literal oracle."""
    w = wl.config(3)
    s = w.snapshot
    gf_ctx.set_snapshot(s.avail, s.sched)
    gf_ctx.set_orders(s.driver_order, s.exec_order)
    apps = _gpu_apps(w.drv, w.exe, w.k)
    gpu = gf_ctx.fit_batch(IND, algo, apps)
    ref = ob.fit_independent(algo, s.avail, ob.make_apps(w.drv, w.exe, w.k), s.driver_order, s.exec_order,
                             closed_form=True)
    _assert_same(gpu, ref, apps)
    sub = wl.Workload(w.name, s, w.drv[:2000], w.exe[:2000], w.k[:2000], w.k_max[:2000], w.flags[:2000])
    _check_properties(algo, sub, gpu)
    # idempotence: the same batch again gives the same bytes (the independent mode must not mutate the snapshot)
    again = gf_ctx.fit_batch(IND, algo, apps)
    assert np.array_equal(again.results, gpu.results) and np.array_equal(again.exec_nodes, gpu.exec_nodes)


def test_device_resident_entry_point_matches_host_entry_point(gf_ctx):
    import torch

    w = wl.config(2)
    s = w.snapshot
    gf_ctx.set_snapshot(s.avail)
    gf_ctx.set_orders(s.driver_order, s.exec_order)
    apps, total_k = gangfit.with_offsets(_gpu_apps(w.drv, w.exe, w.k))
    host = gf_ctx.fit_batch(IND, TIGHT, apps)
    dev = torch.device("cuda:0")
    d_apps = torch.from_numpy(apps.view(np.uint8).copy()).to(dev)
    d_res = torch.zeros(len(apps) * 16, dtype=torch.uint8, device=dev)
    d_exec = torch.zeros(total_k + 1, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    gf_ctx.fit_batch_dev(IND, TIGHT, len(apps), d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), total_k,
                         stream=stream)
    torch.cuda.synchronize()
    res = d_res.cpu().numpy().view(gangfit._native.RESULT_DTYPE)
    ex = d_exec.cpu().numpy().view(np.uint32)[:total_k]
    assert np.array_equal(res, host.results)
    feas = np.nonzero(res["has_capacity"])[0]
    for a in feas:
        o, n = int(apps["exec_off"][a]), int(apps["k"][a])
        assert np.array_equal(ex[o:o + n], host.exec_nodes[o:o + n])


def test_argument_errors(gf_ctx):
    gf_ctx.set_snapshot([[1, 1, 0], [1, 1, 0]])
    with pytest.raises(gangfit.GangfitError) as e:
        gf_ctx.set_orders([0, 1], [0, 0])  # duplicate node in executor order
    assert e.value.code == gangfit._native.GF_ERR_INVALID
    gf_ctx.set_orders([0, 1], [0, 1])
    with pytest.raises(gangfit.GangfitError) as e:
        gf_ctx.fit_batch(IND, TIGHT, _gpu_apps([[0, 0, 0]], [[1, 1, 0]], [-1]))
    assert e.value.code == gangfit._native.GF_ERR_INVALID
    with pytest.raises(gangfit.GangfitError):
        gf_ctx.fit_batch(IND, 7, _gpu_apps([[0, 0, 0]], [[1, 1, 0]], [1]))
    with pytest.raises(gangfit.GangfitError):
        gf_ctx.set_snapshot([[1 << 62, 1, 0]])


def test_recorded_graph_replays_the_same_batches(gf_ctx):
    """gf_graph_begin / end / launch: the recorded *_dev calls replay with the same results as the eager calls."""
    import torch

    from gangfit import workloads as wl

    w = wl.headline(3000, 300)
    s = w.snapshot
    gf_ctx.set_snapshot(s.avail, s.sched)
    gf_ctx.set_orders(s.driver_order, s.exec_order)
    apps, total_k = gangfit.with_offsets(gangfit.make_apps(w.drv, w.exe, w.k))
    dev = torch.device("cuda:0")
    d_apps = torch.from_numpy(apps.view(np.uint8).copy()).to(dev)
    outs = []
    for algo in (TIGHT, EVEN):
        d_res = torch.zeros(len(apps) * 16, dtype=torch.uint8, device=dev)
        d_exec = torch.zeros(total_k + 1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()  # torch fills on ITS stream; the context's stream does not wait for it
        gf_ctx.fit_batch_dev(IND, algo, len(apps), d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), total_k)
        torch.cuda.synchronize()
        want_res, want_exec = d_res.clone(), d_exec.clone()
        gf_ctx.graph_begin()
        for _ in range(4):
            gf_ctx.fit_batch_dev(IND, algo, len(apps), d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), total_k)
        g = gf_ctx.graph_end()
        for _ in range(3):
            d_res.zero_()
            d_exec.zero_()
            torch.cuda.synchronize()
            gf_ctx.graph_launch(g)
            gf_ctx.timer_begin()
            gf_ctx.timer_end()  # waits for the context's stream
            assert torch.equal(d_res, want_res) and torch.equal(d_exec, want_exec)
        gf_ctx.graph_destroy(g)
        outs.append(want_res)
    ref = ob.fit_independent(TIGHT, s.avail, ob.make_apps(w.drv, w.exe, w.k), s.driver_order, s.exec_order, closed_form=True)
    assert np.array_equal(outs[0].cpu().numpy().view(gangfit._native.RESULT_DTYPE), ref.results)
