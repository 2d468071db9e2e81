#!/usr/bin/env python
"""bench.py — gang-fit decisions/sec + FIFO Filter latency on MI355X (BASELINE.json's metric, both halves).

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
  * step      = one pass of the hot path over one batch: the independent gang-fit decision (SparkBinPack + tightlyPackExecutors
                + driver-fit check) of every application of the headline workload's pending table (10 000 nodes x 1 000 pending
                apps, SURVEY.md 8d distributions), inputs already resident in HBM.  The K steps of a window are served in TWO
                regimes and both are reported (`roofline.regimes`):
                  streamed_tickets   K tickets of the resident worker (gf_worker_submit_dev: one launch of fit_worker_kernel
                                     serves the window, `worker_sets` batches in flight; its launch and departure are inside
                                     the window) — what a caller sees that sends batch after batch;
                  launch_per_batch   K launches of fit_independent_kernel on one stream, recorded as one graph
                                     (gf_fit_batch_dev) — what rocprofv3 sees as one dispatch per batch;
                plus what ONE blocking call costs a host that hands over host memory (`blocking_call`, gf_fit_batch) — what the
                reference's only independent-batch consumer issues (unschedulablepods.go:93-166, once a minute).
  * value     = decisions/sec of the whole job = n_gpus * apps_per_batch * K / median window of the FASTER of the two K-step
                regimes; `config.regime` names it and `roofline` describes the kernel of THAT regime (roofline.kernel_ms is that
                kernel's device time per step, HIP events on the stream it runs on: <= ms_per_step).
  * timing    = W warm-up steps, then windows of EXACTLY K steps, each bracketed by barrier + synchronize on both sides,
                wall time = max over ranks of (rank's clock from the opening barrier to its own synchronize after step K).  A
                K-step window is 0.1-0.2 ms at the driver's K = 20, where one scheduler hiccup moves the figure by tens of
                percent, so `--windows` (default 9) such windows are timed and the MEDIAN is reported; every window is listed.
  * N > 1     = the pending-app table shards across ranks (independent decisions: no data-path collective);
                every rank holds the full node table; per-GPU work is fixed -> "weak" scaling.
  * roofline  = ONE definition of `frac` everywhere (headline, config 3, FIFO chain): three measured fractions —
                hbm   = HBM bytes per step (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes of this command, committed
                        under profiles/) / kernel time / 8 TB/s,
                l2    = L2 request bytes (TCC_HIT + TCC_MISS, 128-byte lines) / kernel time / 34.5 TB/s,
                issue = issued instructions (SQ_INSTS_VALU + SALU + LDS + SMEM) / (1 024 SIMDs x 2.4 GHz x kernel time),
                `bound` = the largest of them, `frac` = its value.  The SURVEY.md 8d full-scan formula (which charges bytes the
                lazy scan never reads) and the visited bytes of the in-kernel counters are kept as information, never divided
                by a peak.  `traffic` = the HBM bytes per step of the profile (null when no profile of this kernel is committed).
  * cpu_baseline = the literal C oracle (port of the reference's loops) on the same workload, 1 core; the FIFO Filter
                (`fifo_filter`, the latency half of the metric) carries its own CPU figure: the literal chain on the same
                queue, decision-only and with the per-node efficiency map the reference's SparkBinPack also builds.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # the deployment's setting: before the HIP runtime initialises

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
L2_PEAK_GBPS = 34500.0  # aggregate L2 bandwidth of the eight XCDs (MI355X_MICROARCH.md)
N_SIMDS = 1024          # 256 CUs x 4
CLOCK_GHZ = 2.4
ISSUE_PEAK_GINST = N_SIMDS * CLOCK_GHZ  # one instruction per SIMD and cycle
PROFILE_TAG = "r5an"     # profiles/<tag>_* hold the rocprofv3 passes of this command (tools/profile_round.sh)


def load_profile(name):
    """A committed counter summary (profiles/<name>, written by tools/summarize_profile.py from rocprofv3 --pmc passes of this
    command on the MI355X box), or None."""
    path = os.path.join(REPO, "profiles", name)
    try:
        return json.load(open(path))
    except Exception:
        return None


def three_fractions(time_s, hbm_bytes, l2_bytes, instructions, simds=N_SIMDS, valu_busy=None, profile_time_s=None):
    """The one definition of `frac` (see the module docstring): measured HBM bytes, L2 request bytes and issued instructions of
    one step, each against its own peak over the kernel's time for that step.  Returns (fractions, bound, frac); a quantity
    that was not measured is left out, and bound is None when nothing was."""
    fr = {}
    if hbm_bytes is not None:
        a = hbm_bytes / time_s / 1e9
        fr["hbm"] = {"bytes_per_step": hbm_bytes, "achieved": a, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": a / HBM_PEAK_GBPS}
    if l2_bytes is not None:
        a = l2_bytes / time_s / 1e9
        fr["l2"] = {"bytes_per_step": l2_bytes, "achieved": a, "peak": L2_PEAK_GBPS, "unit": "GB/s", "frac": a / L2_PEAK_GBPS}
    if instructions is not None:
        a = instructions / time_s / 1e9
        peak = simds * CLOCK_GHZ
        fr["issue"] = {"instructions_per_step": instructions, "achieved": a, "peak": peak, "unit": "Ginstr/s", "frac": a / peak,
                       "simds": simds}
    if valu_busy is not None:
        # what the vector ALUs themselves report (rocprofv3's derived VALUBusy of the committed profile: the share of the kernel's
        # cycles in which they process an instruction), carried over to this run's kernel time — the same instructions in less
        # time keep the pipes busier.  The ceiling of a kernel that is neither bytes nor flops: integer / compare / select work.
        f = valu_busy * (profile_time_s / time_s if profile_time_s else 1.0)
        fr["valu"] = {"achieved": f, "peak": 1.0, "unit": "share of cycles the vector ALUs are busy (VALUBusy)", "frac": f,
                      "profile_valu_busy": valu_busy}
    if not fr:
        return fr, None, None
    bound = max(fr, key=lambda k: fr[k]["frac"])
    return fr, bound, fr[bound]["frac"]


def _percentile(xs, q):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, int(round(q * (len(xs) - 1))))]


def _median(xs):
    return _percentile(xs, 0.5)


# ------------------------------------------------------------------------------------------------ CPU baselines (oracle)
# Only these functions touch oracle/: the checker timed as the CPU baseline, never part of the measured GPU path.

def cpu_baseline(w, budget_s: float = 10.0):
    """Literal C oracle (1 thread) on the bench workload: whole batches repeated until ~budget_s of CPU work."""
    from oracle import binding as ob

    s = w.snapshot
    apps = ob.make_apps(w.drv, w.exe, w.k, w.flags)
    ob.fit_independent(0, s.avail, apps[:8], s.driver_order, s.exec_order)  # warm up / build
    n_done, t0 = 0, time.perf_counter()
    while True:
        ob.fit_independent(0, s.avail, apps, s.driver_order, s.exec_order, closed_form=False)
        n_done += len(apps)
        dt = time.perf_counter() - t0
        if dt > budget_s:
            break
    return {
        "value": n_done / dt,
        "unit": "decisions/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{n_done // len(apps)} passes over the same {len(apps)}-app x {len(s.avail)}-node batch, literal C port "
                  f"(oracle/gangfit_oracle.c), 1 thread, {dt:.1f} s",
        "note": "literal C restatement of SparkBinPack+tightlyPackExecutors on dense arrays; the reference's string-keyed maps and "
                "per-call allocations are NOT reproduced: conservative",
    }


def _cpu_worker(args):
    n_nodes, n_apps, budget_s, closed = args
    from gangfit import workloads as wl
    from oracle import binding as ob

    w = wl.headline(n_nodes, n_apps, seed=0x5EED0010)
    s = w.snapshot
    apps = ob.make_apps(w.drv, w.exe, w.k, w.flags)
    ob.fit_independent(0, s.avail, apps[:8], s.driver_order, s.exec_order)
    n_done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        ob.fit_independent(0, s.avail, apps, s.driver_order, s.exec_order, closed_form=closed)
        n_done += len(apps)
    return n_done, time.perf_counter() - t0


def cpu_baseline_variants(n_nodes, n_apps, budget_s: float = 3.0):
    """next to the single-thread literal port, the array-form (closed-form) restatement on one core and a
    'generous' all-cores figure (independent decisions, one process per core)."""
    import multiprocessing as mp

    n_done, dt = _cpu_worker((n_nodes, n_apps, budget_s, True))
    out = {"closed_form_1_core": {"value": n_done / dt, "unit": "decisions/s", "cores": 1, "kind": "port"}}
    try:  # the reference's data-structure shape (string-keyed maps, per-pack efficiency map) on a slice of the batch
        from gangfit import workloads as wl
        from oracle import binding as ob

        wm = wl.headline(n_nodes, n_apps, seed=0x5EED0010)
        sm = wm.snapshot
        am = ob.make_apps(wm.drv, wm.exe, wm.k, wm.flags)[:200]
        t0 = time.perf_counter()
        ob.fit_maps(0, sm.avail, am, sm.driver_order, sm.exec_order, chain=False, sched=sm.sched)
        dtm = time.perf_counter() - t0
        out["reference_shaped_1_core"] = {"value": len(am) / dtm, "unit": "decisions/s", "cores": 1, "kind": "port",
                                          "sample": f"first {len(am)} apps of the batch on string-keyed maps incl. the per-node "
                                                    f"efficiency map of every pack (oracle/gangfit_oracle_maps.cpp), {dtm:.2f} s"}
    except Exception as e:
        out["reference_shaped_1_core"] = {"error": f"{type(e).__name__}: {e}"}
    cores = os.cpu_count() or 1
    try:
        with mp.get_context("spawn").Pool(cores) as pool:
            res = pool.map(_cpu_worker, [(n_nodes, n_apps, budget_s, False)] * cores)
        out["literal_all_cores"] = {"value": sum(r[0] for r in res) / max(r[1] for r in res), "unit": "decisions/s",
                                    "cores": cores, "kind": "port",
                                    "sample": f"{cores} processes x {budget_s:.0f} s of whole-batch passes"}
    except Exception as e:
        out["literal_all_cores"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def cpu_baseline_congested(w, max_apps: int = 48):
    from oracle import binding as ob

    s = w.snapshot
    apps = ob.make_apps(w.drv, w.exe, w.k, w.flags)[:max_apps]
    t0 = time.perf_counter()
    ob.fit_independent(0, s.avail, apps, s.driver_order, s.exec_order, closed_form=False)
    dt = time.perf_counter() - t0
    return {"value": len(apps) / dt, "unit": "decisions/s", "cores": 1, "kind": "port",
            "sample": f"first {len(apps)} apps of the congested batch, literal oracle, {dt:.1f} s"}


def cpu_chain_baseline(algo, avail, sched, zone, driver_order, exec_order, drv, exe, k, flags, reps: int, with_eff_reps: int = 0,
                       maps_reps: int = 0):
    """The FIFO Filter's compute on one CPU core: the literal chain (fitEarlierDrivers + final pack, resource.go:224-262,
    309-328) over the SAME queue, a different head per repetition like the GPU leg.  `with_eff_reps` > 0 adds the
    reference-shaped variant: every successful pack also builds the per-node PackingEfficiencies map (binpack.go:77)."""
    from oracle import binding as ob

    apps = ob.make_apps(drv, exe, k, flags)
    ts = []
    for i in range(reps):
        rolled = np.roll(apps, -i)
        t0 = time.perf_counter()
        ob.fit_fifo_chain(algo, avail, rolled, driver_order, exec_order, sched=sched, zone=zone)
        ts.append((time.perf_counter() - t0) * 1e3)
    out = {"p50_ms": _median(ts), "max_ms": max(ts), "chains": reps, "cores": 1, "kind": "port",
           "sample": f"{reps} chains of {len(apps)} apps over {len(avail)} nodes, literal C restatement on dense arrays, "
                     f"decision only (no efficiency maps, no string-keyed maps)"}
    if with_eff_reps > 0 and sched is not None:
        ts = []
        for i in range(with_eff_reps):
            t0 = time.perf_counter()
            ob.fit_fifo_chain(algo, avail, np.roll(apps, -i), driver_order, exec_order, sched=sched, zone=zone,
                              with_efficiencies=True)
            ts.append((time.perf_counter() - t0) * 1e3)
        out["with_efficiency_maps_p50_ms"] = _median(ts)
        out["with_efficiency_maps_note"] = ("same dense-array chain plus ComputePackingEfficiencies over all nodes on every "
                                            "successful pack (binpack.go:77), which the reference computes and "
                                            "fitEarlierDrivers discards")
    if maps_reps > 0 and sched is not None and algo in (0, 1) and zone is None:
        ts = []
        for i in range(maps_reps):
            t0 = time.perf_counter()
            ob.fit_maps(algo, avail, np.roll(apps, -i), driver_order, exec_order, chain=True, sched=sched)
            ts.append((time.perf_counter() - t0) * 1e3)
        out["reference_shaped_p50_ms"] = _median(ts)
        out["reference_shaped_note"] = ("the same chain on the reference's data structures (oracle/gangfit_oracle_maps.cpp): string "
                                        "node names, hash maps for metadata / reserved / usage, a fresh N-sized reserved map per "
                                        "driver candidate (binpack.go:72) and the per-node efficiency map of every successful pack "
                                        "(binpack.go:77); int64 quantities — still cheaper than resource.Quantity arithmetic")
    return out


# ------------------------------------------------------------------------------------------------ BASELINE config 3

def run_config3(ctx, torch, dev, stream, timed_graph, steps=100):
    """10 000 nodes x 10 000 pending apps, both plain packers, device resident: decisions/s, kernel time, and the roofline as the
    three measured fractions (HBM bytes, L2 request bytes, issued instructions of profiles/pmc_config3.json over this run's
    kernel time).  The visited bytes of the in-kernel counters and the SURVEY.md 8d full-scan formula are listed as
    information only: most visited bytes are served by the L1 / L2, and the formula charges bytes the lazy scan never reads."""
    import gangfit
    from gangfit import workloads as wl

    IND = gangfit.GF_MODE_INDEPENDENT
    w3 = wl.config(3)
    ctx.set_snapshot(w3.snapshot.avail, w3.snapshot.sched)
    ctx.set_orders(w3.snapshot.driver_order, w3.snapshot.exec_order)
    apps3, total_k3 = gangfit.with_offsets(gangfit.make_apps(w3.drv, w3.exe, w3.k, w3.flags))
    d_apps3 = torch.from_numpy(apps3.view(np.uint8).copy()).to(dev)
    d_res3 = torch.zeros(len(apps3) * 16, dtype=torch.uint8, device=dev)
    d_exec3 = torch.zeros(total_k3 + 1, dtype=torch.int32, device=dev)
    alg = wl.algorithmic_bytes(len(w3.snapshot.exec_order), w3.k)
    pmc = load_profile("pmc_config3.json")
    c3 = {}
    for name, algo in (("tightly_pack", gangfit.GF_ALGO_TIGHTLY_PACK), ("distribute_evenly", gangfit.GF_ALGO_DISTRIBUTE_EVENLY)):
        def step3():
            ctx.fit_batch_dev(IND, algo, len(apps3), d_apps3.data_ptr(), d_res3.data_ptr(), d_exec3.data_ptr(), total_k3,
                              stream=stream)
        wall_3, kern_3, _, how3 = timed_graph(step3, steps, 3, 5)
        ctx.scan_stats(enable=True, reset=True)
        step3()
        torch.cuda.synchronize()
        xv, dv = ctx.scan_stats(enable=False, reset=True)
        visited = xv * 24 + dv * 28 + len(apps3) * 88 + 4 * int(w3.k.sum())
        t = (pmc or {}).get(name) or {}
        insts = sum(t.get(c, 0.0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM")) or None
        fr, bound, frac = three_fractions(kern_3 * 1e-3, t.get("hbm_bytes"), t.get("l2_request_bytes"), insts)
        rf = {"kernel": f"fit_independent_kernel<{name}>", "kernel_ms": kern_3, "bound": bound, "frac": frac,
              "achieved": fr[bound]["achieved"] if bound else None, "peak": fr[bound]["peak"] if bound else None,
              "unit": fr[bound]["unit"] if bound else None, "fractions": fr,
              "traffic": t.get("hbm_bytes"),
              "counters_from": (f"profiles/pmc_config3.json ({pmc.get('tag')}): rocprofv3 --pmc passes of `bench.py --config3-only` — a "
                                "committed profile, not this run; the kernel time IS this run's") if pmc else None,
              "wait_fraction": (t["SQ_WAIT_ANY"] / t["SQ_WAVE_CYCLES"]) if t.get("SQ_WAVE_CYCLES") else None,
              "information_only": {"visited_bytes_per_launch": visited, "visited_GBps": visited / (kern_3 * 1e-3) / 1e9,
                                   "algorithmic_bytes_per_launch": alg, "algorithmic_full_scan_GBps": alg / (kern_3 * 1e-3) / 1e9,
                                   "note": "visited = in-kernel counters of this run (mostly served by the L1 / L2: never divided by the "
                                           "HBM peak); algorithmic = SURVEY.md 8d, which charges the full scan the lazy kernel skips"},
              "note": "10 000 wavefronts over 1 024 SIMDs: about ten per SIMD; bound by the depth of the dependent-miss chain and the "
                      "issue slots (DESIGN.md 9), not by bytes"}
        c3[name] = {"decisions_per_s": len(apps3) * steps / wall_3, "kernel_ms": kern_3, "submission": how3, "roofline": rf,
                    "decisions_per_s_by_kernel_time": len(apps3) / (kern_3 * 1e-3), "launches_per_window": steps,
                    "window_is": "one recorded graph of `launches_per_window` launches between synchronises, host clock (rounds 1-4 used 20: "
                                 "the graph's start and the closing synchronise were a tenth of such a window); the kernel is linear in "
                                 "the batch size from ~2 000 applications on, 5.1 us + 0.6 us per 1 000 (profiles/r5n_config3_sizes.txt)"}
    return c3


# ------------------------------------------------------------------------------------------------ the contract line
# The driver parses ONE JSON line from stdout.  Round 4's line carried every leg of the run (26 KB) and could not be parsed;
# the line is now the contract's fields only (< 4 KB, tests/test_bench_line.py) and everything measured goes to
# bench_full.json next to this file (GANGFIT_BENCH_FULL names another path).

LINE_LIMIT = 4096


def _r(x, digits=6):
    """Numbers of the line with `digits` significant digits (the full precision is in bench_full.json)."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    return x


def _trim_roofline(rf):
    """kernel, its time, bound / achieved / peak / unit / frac / traffic, the three fractions, the wait fraction."""
    if not rf:
        return None
    out = {k: _r(rf.get(k)) for k in ("kernel", "kernel_ms", "bound", "achieved", "peak", "unit", "frac", "traffic", "wait_fraction")}
    for k in ("profile_steps", "profile_kernel_us", "profile_frac", "visited_bytes", "visited_over_formula"):
        if rf.get(k) is not None:
            out[k] = _r(rf.get(k), 4)
    fr = rf.get("fractions") or {}
    out["fractions"] = {k: _r(v.get("frac")) for k, v in fr.items()}
    return out


def compact_line(full):
    """The ONE line of the contract from the full result dict: metric, value, unit, n_gpus, steps, warmup, ms_per_step, dtype,
    config, a trimmed roofline (+ the FIFO chain's) and cpu_baseline.  Never longer than LINE_LIMIT characters."""
    line = {k: _r(full.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                        "scaling", "vs_baseline", "dtype", "data")}
    cfg = dict(full.get("config") or {})
    line["config"] = {k: _r(v) for k, v in cfg.items()}
    rf = full.get("roofline") or {}
    t = _trim_roofline(rf) or {}
    reg = rf.get("regimes") or {}
    t["regimes"] = {name: {"value": _r((reg.get(name) or {}).get("value")), "kernel_ms": _r((reg.get(name) or {}).get("kernel_ms"))}
                    for name in ("streamed_tickets", "launch_per_batch") if reg.get(name)}
    bc = reg.get("blocking_call") or {}
    if bc.get("us_per_call") is not None:
        t["regimes"]["blocking_call"] = {"value": _r(bc.get("value")), "us_per_call": _r(bc.get("us_per_call"))}
        t["regimes"]["blocking_call"]["is"] = "SURVEY 8d end-to-end: host memory in and out, one call"
        if bc.get("feasibility_only_us_per_call") is not None:  # gf_fit_feasible: what UnschedulablePodMarker reads
            t["regimes"]["blocking_call"]["feasibility_only_us_per_call"] = _r(bc.get("feasibility_only_us_per_call"))
    cg = (full.get("extras") or {}).get("congested") or {}
    if cg.get("decisions_per_s"):  # the hard case of the same batch: usage ~U[0.95, 1], about half of the gangs do not fit
        t["regimes"]["congested"] = {"value": _r(cg.get("decisions_per_s")), "kernel_ms": _r(cg.get("kernel_ms")),
                                     "feasible_fraction": _r(cg.get("feasible_fraction"), 3), "regime": "launch_per_batch",
                                     "cpu_1_core": _r((cg.get("cpu_baseline") or {}).get("value"))}
    fc = rf.get("fifo_chain")
    if fc:
        c = _trim_roofline(fc)
        for k in ("filter_p50_ms", "filter_p99_ms", "filter_warm_p50_ms", "lone_wavefront_issue_frac", "kernel_ms_le_filter_p50"):
            c[k] = _r(fc.get(k))
        t["fifo_chain"] = c
    line["roofline"] = t
    cb = full.get("cpu_baseline")
    if cb:
        c = {k: _r(cb.get(k)) for k in ("value", "unit", "cores", "kind")}
        c["sample"] = str(cb.get("sample") or "")[:160]
        f = cb.get("fifo_chain") or {}
        if f:
            c["fifo_chain"] = {k: _r(f.get(k)) for k in ("literal_p50_ms", "with_efficiency_maps_p50_ms", "reference_shaped_p50_ms",
                                                         "gpu_cold_p50_ms", "speedup_cold_p50", "cores", "kind")}
        ac = (full.get("cpu_baseline_variants") or {}).get("literal_all_cores") or {}
        if ac.get("value"):
            c["all_cores"] = {"value": _r(ac.get("value")), "cores": ac.get("cores")}
        line["cpu_baseline"] = c
    line["full"] = full.get("full_results_file")
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT and "fifo_chain" in line["roofline"]:
        line["roofline"]["fifo_chain"].pop("fractions", None)
        text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:  # cannot happen with the fields above; a guard, so that the driver always gets a line it can parse
        for k in ("sample",):
            if "cpu_baseline" in line:
                line["cpu_baseline"].pop(k, None)
        line["config"] = {k: v for k, v in line["config"].items() if not isinstance(v, str) or len(v) < 80}
        text = json.dumps(line, separators=(",", ":"))
    return text


def emit(full):
    """Write everything to bench_full.json, print the contract line (the LAST thing on stdout)."""
    path = os.environ.get("GANGFIT_BENCH_FULL", os.path.join(REPO, "bench_full.json"))
    try:
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        full["full_results_file"] = os.path.relpath(path, REPO) if path.startswith(REPO) else path
    except OSError as e:
        full["full_results_file"] = f"not written: {e}"
    sys.stdout.flush()
    print(compact_line(full), flush=True)


def self_launch(n_gpus):
    """`python bench.py --gpus N` with N > 1 and no rank environment: start N ranks of this same command, one per GPU, the way
    the driver would (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P).
    Returns the exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------ main

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)  # ~15 ms per window: the closing barrier of an N-GPU run stays below 1 %
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--windows", type=int, default=9, help="timed windows of exactly --steps steps; the median is reported")
    ap.add_argument("--worker-sets", type=int, default=-1,
                    help="groups of wavefronts of the resident worker = independent batches in flight on the device; -1 = the library's "
                         "choice (three applications per wavefront, as many sets as fit: 11 sets of 21 workgroups for 1 000 "
                         "applications), 0 = launch path only")
    ap.add_argument("--nodes", type=int, default=10000)
    ap.add_argument("--apps", type=int, default=1000)
    ap.add_argument("--filter-calls", type=int, default=1000, help="FIFO Filter calls (different heads) behind p50/p99")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip everything but the headline, its roofline and the FIFO Filter")
    ap.add_argument("--fifo-protocols", default="cold,warm,retry",
                    help="which FIFO Filter protocols run (the profiling passes use `cold`: every fit_fifo_solo_kernel launch of "
                         "the trace is then a full replay of the headline chain)")
    ap.add_argument("--config3-only", action="store_true",
                    help="nothing but BASELINE config 3 (10 000 nodes x 10 000 apps, both plain packers) and its roofline: the "
                         "counter passes of tools/profile_round.sh use it")
    ap.add_argument("--headline-only", action="store_true",
                    help="nothing but the headline batch and its roofline (the profiling passes use it: every launch of "
                         "fit_independent_kernel in the trace is then a headline launch)")
    args = ap.parse_args()
    if args.headline_only:
        args.no_extras = args.no_cpu_baseline = True
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    import torch

    import gangfit
    from gangfit import workloads as wl

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start `python bench.py --gpus N` without WORLD_SIZE in the "
                         "environment (it launches its own ranks) or under torch.distributed.run with --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path is the only product path (no CPU fallback)")
    n_visible = torch.cuda.device_count()
    dev_index = local_rank % max(1, n_visible)  # one rank per GPU; ranks only share a device in the one-GPU smoke run
    torch.cuda.set_device(dev_index)
    dist = None
    backend = "none"
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" (= RCCL over xGMI) whenever every rank has a GPU of its own.  RCCL refuses two ranks on one device, so the
        # smoke run of this N > 1 control flow on a ONE-GPU box (tools/smoke_two_ranks_one_gpu.sh) runs over gloo: chosen
        # by itself when the box has fewer GPUs than ranks, or by GANGFIT_BENCH_BACKEND=gloo.  The line says which it was.
        backend = os.environ.get("GANGFIT_BENCH_BACKEND", "nccl" if n_visible >= world else "gloo")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", dev_index)
    local_rank = dev_index

    # ---- workload: same node table everywhere, a different slice of the pending queue per rank
    w = wl.headline(args.nodes, args.apps, seed=0x5EED0010 + 7919 * rank)
    base = wl.headline(args.nodes, args.apps, seed=0x5EED0010)
    w.snapshot = base.snapshot
    s = w.snapshot
    ctx = gangfit.Context(local_rank)
    ctx.set_snapshot(s.avail, s.sched)
    ctx.set_orders(s.driver_order, s.exec_order)
    apps, total_k = gangfit.with_offsets(gangfit.make_apps(w.drv, w.exe, w.k, w.flags))
    d_apps = torch.from_numpy(apps.view(np.uint8).copy()).to(dev)
    d_res = torch.zeros(len(apps) * 16, dtype=torch.uint8, device=dev)
    d_exec = torch.zeros(total_k + 1, dtype=torch.int32, device=dev)
    stream = 0  # 0 = the context's own (non-blocking) stream: kernels, timer events and recorded graphs all live on it
    IND, FIFO = gangfit.GF_MODE_INDEPENDENT, gangfit.GF_MODE_FIFO_CHAIN
    TIGHT, EVEN = gangfit.GF_ALGO_TIGHTLY_PACK, gangfit.GF_ALGO_DISTRIBUTE_EVENLY

    def step(algo=TIGHT):
        ctx.fit_batch_dev(IND, algo, len(apps), d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), total_k,
                          stream=stream)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def window(fn, steps):
        """EXACTLY `steps` calls of fn between barrier + synchronize on both sides; (max-over-ranks wall s, kernel ms/step)."""
        barrier()
        t0 = time.perf_counter()
        ctx.timer_begin(stream)
        for _ in range(steps):
            fn()
        ev_ms = ctx.timer_end()  # HIP events on the launch stream; returns when the last kernel is done
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0  # this rank's K steps, device idle again; the slowest rank defines the window (MAX below)
        barrier()                        # closes the bracket; its own latency (an RCCL barrier on N > 1) is not work of the K steps
        if dist is not None:
            t = torch.tensor([wall], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        return wall, ev_ms / steps

    def timed(fn, steps, warmup, windows):
        for _ in range(warmup):
            fn()
        ws = [window(fn, steps) for _ in range(max(1, windows))]
        walls = [a for a, _ in ws]
        return _median(walls), _median([b for _, b in ws]), walls

    def timed_graph(fn, steps, warmup, windows):
        """timed(), with the K steps of a window submitted as one recorded graph (single rank; falls back to eager submission)."""
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        g = None
        try:
            ctx.graph_begin(stream)
            for _ in range(steps):
                fn()
            g = ctx.graph_end(stream)
        except Exception:
            g = None
        if g is None or dist is not None:
            if g is not None:
                ctx.graph_destroy(g)
            return timed(fn, steps, 0, windows) + ("eager",)
        walls_, kerns_ = [], []
        for i in range(max(1, windows) + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.timer_begin(stream)
            ctx.graph_launch(g, stream)
            ev = ctx.timer_end()
            torch.cuda.synchronize()
            if i:
                walls_.append(time.perf_counter() - t0)
                kerns_.append(ev / steps)
        ctx.graph_destroy(g)
        return _median(walls_), _median(kerns_), walls_, "graph"

    if args.config3_only:
        c3 = run_config3(ctx, torch, dev, stream, timed_graph)
        ctx.close()
        print(json.dumps({"config3_10k_nodes_x_10k_apps": c3}))
        return

    # The K steps of a window are submitted as ONE recorded graph (gf_graph_*: K kernel nodes, the same launches the eager
    # calls make): a 1 000-application batch takes about as long on the device as the host needs to submit one kernel, so
    # eager submission measures the host.  The eager figure (one gf_fit_batch_dev call per step) is reported next to it.
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    graph = None
    try:
        ctx.graph_begin(stream)
        for _ in range(args.steps):
            step()
        graph = ctx.graph_end(stream)
    except Exception:
        graph = None
    if dist is not None:  # every rank must take the same path below (the windows contain collectives)
        okf = torch.tensor([1 if graph is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        if int(okf.item()) == 0 and graph is not None:
            ctx.graph_destroy(graph)
            graph = None
    eager_wall, eager_kern_ms, eager_walls = timed(step, args.steps, 0, 3 if graph is not None else args.windows)
    if graph is not None:
        def window_graph():
            barrier()
            t0 = time.perf_counter()
            ctx.timer_begin(stream)
            ctx.graph_launch(graph, stream)  # exactly K recorded steps
            ev_ms = ctx.timer_end()
            torch.cuda.synchronize()
            wall_ = time.perf_counter() - t0
            barrier()
            if dist is not None:
                t = torch.tensor([wall_], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                wall_ = float(t.item())
            return wall_, ev_ms / args.steps

        window_graph()
        ws = [window_graph() for _ in range(max(1, args.windows))]
        walls = [a for a, _ in ws]
        wall, kern_ms = _median(walls), _median([b for _, b in ws])
    else:
        wall, kern_ms, walls = eager_wall, eager_kern_ms, eager_walls
    # ---- the same K steps through the RESIDENT WORKER (gf_worker_*, gangfit_worker.inc): the batches of an independent-mode
    #      window do not depend on each other, so nothing but the launch model makes them wait for each other.  A window posts
    #      its K batches as K tickets (one doorbell), the worker — ONE launch, `worker_sets` groups of wavefronts taking the
    #      tickets in turn — serves them with several in flight and leaves when the last one is done (gf_worker_stop), so the
    #      closing synchronize finds an idle device.  The worker's launch and its departure are inside the window.  Every ticket
    #      writes its own result / placement arrays (eight sets in rotation); they are compared with the launch path's.
    seq_wall, seq_walls = wall, walls
    worker_info = None
    used_worker = False
    wkern = []
    worker_sets_used = None
    if args.worker_sets != 0:
        NOUT = 8
        w_res = [torch.zeros_like(d_res) for _ in range(NOUT)]
        w_exec = [torch.zeros_like(d_exec) for _ in range(NOUT)]
        try:
            ctx.set_option("worker_sets", max(0, args.worker_sets))  # 0 = the library chooses at every launch
            arr = ctx.worker_batches([(len(apps), d_apps.data_ptr(), w_res[i % NOUT].data_ptr(), w_exec[i % NOUT].data_ptr(), total_k)
                                      for i in range(args.steps)], leave_after=True)  # a window is a bounded stream: K tickets, then the worker leaves

            def window_worker():
                barrier()
                t0 = time.perf_counter()
                ctx.worker_submit_prepared(TIGHT, arr)  # exactly K tickets
                ctx.worker_stop()                       # served, then the worker leaves the device
                torch.cuda.synchronize()
                wall_ = time.perf_counter() - t0
                try:  # HIP events on the worker's own stream around its launch: device time of this window's K tickets
                    kms, ktk = ctx.worker_kernel_time()
                    if ktk == args.steps:
                        wkern.append(kms / ktk)
                except Exception:
                    pass
                barrier()
                if dist is not None:
                    t = torch.tensor([wall_], dtype=torch.float64, device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    wall_ = float(t.item())
                return wall_

            wkern = []
            for _ in range(3):
                window_worker()
            del wkern[:]
            wwalls = [window_worker() for _ in range(max(1, args.windows))]
            step(TIGHT)  # the launch path's answer into d_res / d_exec
            torch.cuda.synchronize()
            same = all(bool(torch.equal(w_res[i], d_res)) and bool(torch.equal(w_exec[i], d_exec)) for i in range(min(NOUT, args.steps)))
            try:
                worker_sets_used, worker_blocks_used = ctx.worker_geometry()
            except Exception:
                worker_sets_used, worker_blocks_used = (args.worker_sets if args.worker_sets > 0 else None), None
            worker_info = {"sets": worker_sets_used, "workgroups_per_set": worker_blocks_used,
                           "window_ms": [x * 1e3 for x in wwalls], "answers_equal_launch_path": same,
                           "kernel_ms_per_ticket": _median(wkern) if wkern else None, "stats": ctx.worker_stats()}
            if dist is not None:
                okf = torch.tensor([1 if same else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(okf, op=dist.ReduceOp.MIN)
                same = bool(int(okf.item()))
            if same and _median(wwalls) < wall:
                wall, walls, used_worker = _median(wwalls), wwalls, True
        except Exception as e:
            worker_info = {"error": f"{type(e).__name__}: {e}"}
    decisions_per_s = world * len(apps) * args.steps / wall

    # ---- roofline of the kernel behind `value` (and, beside it, of the other regime's kernel)
    alg_bytes = wl.algorithmic_bytes(len(s.exec_order), w.k)  # full scan of the executor order per decision
    ctx.scan_stats(enable=True, reset=True)
    step(TIGHT)
    torch.cuda.synchronize()
    xvis, dvis = ctx.scan_stats(enable=False, reset=True)
    # what the lazy scan touches: 24 B per executor slot evaluated, 28 B per driver position (24 B + its 4-byte node id),
    # the 64-byte app record + 16-byte result + 8 B of index words, 4 B per placement written
    visited_bytes = xvis * 24 + dvis * 28 + len(apps) * 88 + 4 * int(w.k.sum())
    visited_worker = None
    if used_worker:
        try:  # the same counters inside the worker, per ticket.  A SHORT stream: the counters are two device-wide atomics per
            # application, a ticket then takes twenty times as long, and a long stream of such tickets lets the leader idle out
            # between two postings (round 6's K = 2 000 profile held 31 relaunches of 64 tickets each from this one window)
            n_cnt = min(args.steps, 20)
            arr_cnt = ctx.worker_batches([(len(apps), d_apps.data_ptr(), w_res[i % NOUT].data_ptr(), w_exec[i % NOUT].data_ptr(), total_k)
                                          for i in range(n_cnt)], leave_after=True)
            ctx.scan_stats(enable=True, reset=True)
            ctx.worker_submit_prepared(TIGHT, arr_cnt)
            ctx.worker_stop()
            torch.cuda.synchronize()
            xw, dw = ctx.scan_stats(enable=False, reset=True)
            visited_worker = (xw * 24 + dw * 28) / n_cnt + len(apps) * 88 + 4 * int(w.k.sum())
        except Exception:
            visited_worker = None
    try:
        floor_us = ctx.launch_floor(stream, 400)
    except Exception:
        floor_us = None
    try:
        read_peak, copy_peak = ctx.hbm_probe(2 << 30, 10)
    except Exception:
        read_peak = copy_peak = None
    # one launch at a time between its own pair of events on an idle stream: what a LONE batch costs (the launch is not
    # overlapped with a predecessor).  kern_ms above — the K-step window divided by K — is what rocprofv3's average dispatch
    # duration agrees with (profiles/<tag>_summary.md, headline-only table).
    singles = []
    for _ in range(60):
        ctx.timer_begin(stream)
        step(TIGHT)
        singles.append(ctx.timer_end())
    isolated_launch_ms = _median(singles[10:])
    prof_all = load_profile("pmc_headline.json") or {}
    # one entry per profiled command (tools/profile_round.sh: the driver's --steps 20 --warmup 5 and bench.py's defaults); the
    # counters per step depend on K (launch ramp, idle polling), so this run reads the entry taken at ITS --steps — or, failing
    # that, the nearest one, and says so
    prof_runs = prof_all.get("runs") or {}
    prof_key = f"steps{args.steps}"
    if prof_runs and prof_key not in prof_runs:
        prof_key = min(prof_runs, key=lambda k: abs(int(k[5:]) - args.steps))
    prof = dict(prof_runs.get(prof_key) or prof_all)
    prof.setdefault("tag", prof_all.get("tag"))
    prof_steps = prof.get("steps")

    def kernel_roofline(kernel, regime, kernel_ms_step, how, per_step, visited):
        """per_step: the profile's counters of one step of this kernel ({hbm_bytes, l2_request_bytes, instructions, ...}) or None."""
        ps = per_step or {}
        own_ns = ps.get("rocprof_ns_per_ticket") or ps.get("rocprof_median_dispatch_ns") or ps.get("rocprof_avg_dispatch_ns")
        fr, bound, frac = three_fractions(kernel_ms_step * 1e-3, ps.get("hbm_bytes"), ps.get("l2_request_bytes"), ps.get("instructions"),
                                          valu_busy=ps.get("valu_busy"), profile_time_s=(own_ns * 1e-9) if own_ns else None)
        return {"kernel": kernel, "regime": regime, "kernel_ms": kernel_ms_step, "kernel_ms_is": how,
                "bound": bound, "frac": frac,
                "achieved": fr[bound]["achieved"] if bound else None, "peak": fr[bound]["peak"] if bound else None,
                "unit": fr[bound]["unit"] if bound else None, "fractions": fr,
                "traffic": ps.get("hbm_bytes"),
                "counters_from": (f"profiles/pmc_headline.json ({prof.get('tag')}, entry runs.{prof_key}: `{prof.get('command')}`): separate "
                                  "rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; TCC_HIT + TCC_MISS x 128 B; "
                                  "SQ_INSTS_*), per step — a committed profile, not this run; the kernel time IS this run's") if per_step else
                                 "no committed counter profile of this kernel: fractions unmeasured",
                "profile_steps": prof_steps, "profile_steps_match": (prof_steps == args.steps) if per_step else None,
                # the same counters over the PROFILE's own kernel time (rocprofv3's median dispatch / K): what a reader of
                # profiles/pmc_headline.json reproduces without this run
                "profile_kernel_us": (ps.get("rocprof_ns_per_ticket") or ps.get("rocprof_median_dispatch_ns") or
                                      ps.get("rocprof_avg_dispatch_ns") or 0) * 1e-3 or None,
                "profile_frac": ((ps.get("fractions_over_own_duration") or {}).get(bound) if bound else None),
                "visited_bytes": visited,
                "visited_over_formula": (visited / alg_bytes) if visited else None,
                "wait_fraction": ps.get("wait_fraction"),
                "information_only": {
                    "visited_bytes_per_step": visited, "visited_GBps": (visited / (kernel_ms_step * 1e-3) / 1e9) if visited else None,
                    "algorithmic_bytes_per_step": alg_bytes, "algorithmic_full_scan_GBps": alg_bytes / (kernel_ms_step * 1e-3) / 1e9,
                    "note": "visited = in-kernel counters of this run (the scan is lazy like the reference's loop; the 240 KB table is "
                            "cache resident, so these bytes are mostly L1 / L2 hits and are never divided by the HBM peak); "
                            "algorithmic = SURVEY.md 8d's full-scan formula, which charges bytes nobody reads"}}

    launch_rf = kernel_roofline("fit_independent_kernel<tightly-pack>", "launch_per_batch", kern_ms,
                                "HIP events on the launch stream around a K-step window / K (what rocprofv3's average dispatch "
                                "duration agrees with)", prof.get("launch_path"), visited_bytes)
    worker_rf = None
    if worker_info and "error" not in worker_info:
        wk_ms = worker_info.get("kernel_ms_per_ticket") or (_median(worker_info["window_ms"]) / args.steps)
        worker_rf = kernel_roofline("fit_worker_kernel<tightly-pack>", "streamed_tickets", wk_ms,
                                    "HIP events on the worker's stream around its ONE dispatch per window / the K tickets it served "
                                    "(gf_worker_kernel_time)" if worker_info.get("kernel_ms_per_ticket") else
                                    "window wall time / K (the worker's events were not available)",
                                    prof.get("worker"), visited_worker)
        worker_rf["issue_note"] = ("the worker's instruction count includes the polling of 3 x 64 workgroups that wait for their next "
                                   "ticket: issue slots taken, not decisions made")
    roofline = dict(worker_rf if used_worker else launch_rf)
    roofline["other_regime"] = launch_rf if used_worker else worker_rf
    e2e_ref = {}
    roofline["regimes"] = {
        "streamed_tickets": ({"value": world * len(apps) * args.steps / _median(worker_info["window_ms"]) * 1e3,
                              "ms_per_step": _median(worker_info["window_ms"]) / args.steps, "kernel": "fit_worker_kernel<tightly-pack>",
                              "kernel_ms": worker_rf["kernel_ms"], "batches_in_flight": worker_sets_used,
                              "what": "K tickets per window through the resident worker, its launch and departure inside the window"}
                             if worker_rf else None),
        "launch_per_batch": {"value": world * len(apps) * args.steps / seq_wall, "ms_per_step": seq_wall / args.steps * 1e3,
                             "kernel": "fit_independent_kernel<tightly-pack>", "kernel_ms": kern_ms,
                             "isolated_launch_ms": isolated_launch_ms,
                             "what": "K launches on one stream, recorded as one graph (rounds 1-2 reported this as value); "
                                     "isolated_launch_ms = events around ONE launch on an idle stream"},
        "blocking_call": e2e_ref,  # filled below (end_to_end): one gf_fit_batch call with host memory in and out
        "value_is": "streamed_tickets" if used_worker else "launch_per_batch",
        "reference_call_site": "the reference's only independent-batch consumer (unschedulablepods.go:77-129) issues ONE batch per "
                               "minute: it sees `blocking_call`, not the streamed rate"}
    roofline["launch_floor_us"] = floor_us
    roofline["frac_of_launch_floor"] = (floor_us / (kern_ms * 1e3)) if floor_us else None
    roofline["measured_read_stream_GBps"] = read_peak
    roofline["measured_copy_GBps"] = copy_peak
    roofline["note"] = ("10 000 nodes x 24 B = 240 KB of table: cache resident after first touch, so no byte count comes near a "
                        "bandwidth roof; a batch is ~1 000 wavefronts on 1 024 SIMDs, each a short chain of dependent L2 / HBM round "
                        "trips (DESIGN.md 4.1).  frac_of_launch_floor = empty-kernel launch / the launch path's kernel.")

    out = {
        "metric": "gang-fit decisions/sec at 10k nodes x 1k pending apps (+ p99 Filter latency: fifo_filter)",
        "value": decisions_per_s,
        "unit": "decisions/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int64",
        "data": "synthetic",
        "config": {"workload": f"independent batch, tightly-pack, {args.nodes} nodes x {args.apps} pending apps per GPU, "
                               "3-D (cpu milli, mem bytes, gpu) int64, SURVEY.md 8d/C2 distributions, seed 0x5EED0010",
                   "nodes": args.nodes, "apps_per_gpu": args.apps, "algo": "tightly-pack", "mode": "independent",
                   "batches_in_flight": worker_sets_used if used_worker else 1,
                   "regime": "streamed_tickets (resident worker)" if used_worker else "launch_per_batch (one graph of K launches)",
                   "sharding": "pending apps across ranks, node table replicated, no collective",
                   # what the process group was: "nccl" = RCCL over xGMI with one rank per GPU (rccl_ranks = its world size);
                   # "gloo" only in the one-GPU smoke run of the N > 1 control flow; "none" for N = 1
                   "backend": backend, "rccl_ranks": (dist.get_world_size() if (dist is not None and backend == "nccl") else 0)},
        "timing": {"windows": len(walls), "steps_per_window": args.steps, "statistic": "median of the windows (max over ranks each)",
                   "submission": ("resident worker: K tickets per window, one doorbell; the worker's launch and its departure are "
                                  "inside the window (gf_worker_submit_dev + gf_worker_stop)" if used_worker else
                                  "one recorded graph of K steps per window (gf_graph_*: K kernel nodes)" if graph is not None
                                  else "eager: one gf_fit_batch_dev call per step"),
                   "resident_worker": worker_info,
                   "launch_path": {"value": world * len(apps) * args.steps / seq_wall, "ms_per_step": seq_wall / args.steps * 1e3,
                                   "window_ms": [x * 1e3 for x in seq_walls],
                                   "note": "the K kernel nodes of a window one after the other on one stream (what rounds 1-2 "
                                           "reported as value; roofline.kernel_ms comes from these windows)"},
                   "window_ms": [x * 1e3 for x in walls], "best_ms_per_step": min(walls) / args.steps * 1e3,
                   "worst_ms_per_step": max(walls) / args.steps * 1e3,
                   "eager_ms_per_step": eager_wall / args.steps * 1e3, "eager_kernel_ms": eager_kern_ms,
                   "eager_note": "the same K steps submitted by K gf_fit_batch_dev calls from Python (host-bound)"},
        "roofline": roofline,
    }

    def fifo_latency(c, algo, queue, calls, warm=5):
        """One blocking gf_fit_batch(FIFO_CHAIN) per call, a different head each time.  The interpreter's cyclic garbage
        collector is held off while the calls are timed (a generation-2 pass over this process's arrays costs tens of
        milliseconds and belongs to Python, not to the Filter)."""
        import gc

        lat, o = [], None
        rolled_all = [np.roll(queue, -i) for i in range(calls + warm)]
        gc.collect()
        gc.disable()
        try:
            for i in range(calls + warm):
                t0 = time.perf_counter()
                o = c.fit_batch(FIFO, algo, rolled_all[i])
                dt = time.perf_counter() - t0
                if i >= warm:
                    lat.append(dt * 1e3)
        finally:
            gc.enable()
        return lat, o

    if rank == 0 and world == 1 and not args.headline_only:
        # ---- the same batch through the host entry point (what a cgo caller pays): app records in host memory in, results and
        #      placements in host memory out.  Called through the raw ctypes symbol with preallocated buffers — the numpy
        #      marshalling of gangfit.Context.fit_batch costs more than the call itself and is not part of the library.
        try:
            import ctypes as C

            from gangfit import _native as N

            happs, htotal = gangfit.with_offsets(gangfit.make_apps(w.drv, w.exe, w.k, w.flags))
            hres = np.zeros(len(happs), dtype=N.RESULT_DTYPE)
            hexec = np.zeros(htotal + 1, dtype=np.uint32)
            lib, h = ctx._lib, ctx._h
            pa, pr, pe = N.ptr(happs), N.ptr(hres), N.ptr(hexec)

            def host_batch():
                rc = lib.gf_fit_batch(h, IND, TIGHT, len(happs), pa, pr, pe, htotal, None)
                if rc != 0:
                    raise RuntimeError(f"gf_fit_batch: {rc}")

            e2e_steps = max(20, min(args.steps, 200))
            e2e_wall, _, e2e_walls = timed(host_batch, e2e_steps, 5, min(args.windows, 5))
            dres = d_res.cpu().numpy().view(N.RESULT_DTYPE)
            # the same blocking host batch through the resident worker (gf_worker_fit: records written into a pinned slice the
            # device reads in place, answers written by the device into pinned memory; no launch, no stream operation)
            wk = None
            try:
                wres = np.zeros(len(happs), dtype=N.RESULT_DTYPE)
                wexec = np.zeros(htotal + 1, dtype=np.uint32)
                pwr, pwe = N.ptr(wres), N.ptr(wexec)

                def worker_batch():
                    rc = lib.gf_worker_fit(h, TIGHT, len(happs), pa, pwr, pwe, htotal)
                    if rc != 0:
                        raise RuntimeError(f"gf_worker_fit: {rc}")

                def lat(fn, calls=300):
                    for _ in range(20):
                        fn()
                    ts = []
                    for _ in range(calls):
                        t0 = time.perf_counter()
                        fn()
                        ts.append(time.perf_counter() - t0)
                    return _median(ts)

                w_med = lat(worker_batch)
                l_med = lat(host_batch)
                try:  # where the last blocking call spent its time (host clock, gf_call_phases)
                    phases = ctx.call_phases()
                except Exception:
                    phases = None
                # the same 1 000 applications through gf_fit_feasible: HasCapacity only (what DoesPodExceedClusterCapacity
                # reads, unschedulablepods.go:132-166), one byte per application back, the answers watched as they arrive
                hfeas = np.zeros(len(happs), dtype=np.uint8)
                pf = N.ptr(hfeas)

                def feasible_batch():
                    if lib.gf_fit_feasible(h, TIGHT, len(happs), pa, pf) != 0:
                        raise RuntimeError("gf_fit_feasible")

                f_med = lat(feasible_batch)
                try:
                    f_phases = ctx.call_phases()
                except Exception:
                    f_phases = None
                host_batch()
                f_same = bool(np.array_equal(hfeas.astype(bool), hres["has_capacity"].astype(bool)))
                # ... and ONE application per call (DoesPodExceedClusterCapacity for a single pod, gf_spark_binpack)
                k1 = int(happs[0]["k"])

                def worker_one():
                    if lib.gf_worker_fit(h, TIGHT, 1, pa, pwr, pwe, k1) != 0:
                        raise RuntimeError("gf_worker_fit")

                def launch_one():
                    if lib.gf_fit_batch(h, IND, TIGHT, 1, pa, pr, pe, k1, None) != 0:
                        raise RuntimeError("gf_fit_batch")

                w_one = lat(worker_one)
                l_one = lat(launch_one)
                worker_batch()
                host_batch()
                ctx.worker_stop()
                wk = {"ms_per_batch": w_med * 1e3, "decisions_per_s": len(happs) / w_med,
                      "gf_fit_batch_ms_per_batch_same_protocol": l_med * 1e3,
                      "gf_fit_batch_phases_us": phases,
                      "gf_fit_feasible_us_per_batch": f_med * 1e6, "gf_fit_feasible_phases_us": f_phases,
                      "gf_fit_feasible_equals_has_capacity": f_same,
                      "one_application_per_call_us": {"gf_worker_fit": w_one * 1e6, "gf_fit_batch": l_one * 1e6},
                      "results_equal": bool(np.array_equal(wres, hres) and np.array_equal(wexec, hexec)),
                      "protocol": "median of 300 blocking calls, one after the other, worker resident",
                      "note": "a lone 1 000-application batch on the worker is a relay over the host link and back — no faster "
                              "than a launch; a lone SMALL batch is (no dispatch, no completion interrupt); the worker's "
                              "throughput shows with tickets in flight (timing.resident_worker)"}
            except Exception as e:
                wk = {"error": f"{type(e).__name__}: {e}"}
            out["end_to_end"] = {
                "through_the_resident_worker": wk,
                "decisions_per_s": len(happs) * e2e_steps / e2e_wall, "ms_per_batch": e2e_wall / e2e_steps * 1e3,
                "entry_point": "gf_fit_batch: app records in host memory in, results + placements in host memory out (64 KB + ~64 KB "
                               "per batch, read / written by the kernel through pinned staging buffers), snapshot resident; blocking",
                "windows": len(e2e_walls), "results_equal_device_resident_path": bool(np.array_equal(hres, dres)) if rank == 0 and world == 1 else None}
            wkd = wk if isinstance(wk, dict) and "error" not in wk else {}
            e2e_ref.update({"entry_point": "gf_fit_batch (host memory in / out, blocking), one call after the other",
                            "us_per_call": (wkd.get("gf_fit_batch_ms_per_batch_same_protocol") or e2e_wall / e2e_steps * 1e3) * 1e3,
                            "value": len(happs) / ((wkd.get("gf_fit_batch_ms_per_batch_same_protocol") or e2e_wall / e2e_steps * 1e3) * 1e-3),
                            "phases_us": wkd.get("gf_fit_batch_phases_us"),
                            "feasibility_only_us_per_call": wkd.get("gf_fit_feasible_us_per_batch"),
                            "feasibility_only_phases_us": wkd.get("gf_fit_feasible_phases_us"),
                            "feasibility_only_equals_has_capacity": wkd.get("gf_fit_feasible_equals_has_capacity"),
                            "feasibility_only_is": "gf_fit_feasible: HasCapacity of the same 1 000 applications and nothing else (what "
                                                   "UnschedulablePodMarker reads); the kernel's collecting workgroup writes one byte per "
                                                   "application to pinned memory and the caller watches them arrive instead of waiting "
                                                   "for the stream",
                            "phases_are": "host clock: stage = validate + copy 1 000 records into pinned memory; launch = the launch call; "
                                          "wait = the stream (dispatch, the kernel reading the records from and writing the answers to "
                                          "pinned host memory over the host link, its completion signal); copy_out = 64 KB to the "
                                          "caller's arrays.  Round 4 measured a kernel that announces its own completion in pinned "
                                          "memory instead (write-through answers + a polled word): 30.3 us per call against 23.3 — removed",
                            "through_gf_worker_fit_us_per_call": (wkd.get("ms_per_batch") * 1e3) if wkd.get("ms_per_batch") else None,
                            "one_application_per_call_us": wkd.get("one_application_per_call_us")})
        except Exception as e:
            out["end_to_end"] = {"error": f"{type(e).__name__}: {e}"}
        # ---- the latency half of the metric: FIFO Filter = chain of (apps-1) earlier drivers + the filtered one.
        #      Three protocols through the raw C symbol (preallocated buffers; the numpy marshalling of Context.fit_batch is
        #      comparable to a resumed chain):
        #        cold   a different head (rotation of the queue) per call: no two queues share a prefix, every chain replays
        #               all earlier drivers like the reference does (resource.go:309-328) — rounds 1 and 2 measured this;
        #        warm   creation-order heads on the unchanged snapshot: the Filter of driver j follows the Filter of driver
        #               j - 1 (which did not get a reservation: that is why a thousand drivers are pending), j cycling over the
        #               last 256 positions of the queue — the chain resumes from the last checkpoint of the common prefix;
        #        retry  the same head again (kube-scheduler retrying one pod).
        try:
            import ctypes as C
            import gc

            from gangfit import _native as N

            lib, h = ctx._lib, ctx._h
            fq, ftotal = gangfit.with_offsets(apps)
            n_q = len(fq)
            fres = np.zeros(n_q, dtype=N.RESULT_DTYPE)
            fexec = np.zeros(ftotal + 1, dtype=np.uint32)
            ffailed = C.c_int32(-1)
            pr_, pe_, pf_ = N.ptr(fres), N.ptr(fexec), C.byref(ffailed)

            def chain_call(ptr_apps, n):
                t0 = time.perf_counter()
                rc = lib.gf_fit_batch(h, FIFO, TIGHT, n, ptr_apps, pr_, pe_, ftotal + 1, pf_)
                dt = time.perf_counter() - t0
                if rc != 0:
                    raise RuntimeError(f"gf_fit_batch(FIFO): {rc}")
                return dt * 1e3

            def protocol(queues):  # [(array kept alive, pointer, n)]
                gc.collect()
                gc.disable()
                try:
                    ctx.chain_cache_stats(reset=True)
                    lat_ = [chain_call(p_, n_) for _, p_, n_ in queues]
                    st_ = ctx.chain_cache_stats(reset=True)
                finally:
                    gc.enable()
                return lat_, st_

            def summary(lat_, st_, skip):
                lat_ = lat_[skip:]
                return {"p50_ms": _percentile(lat_, 0.5), "p99_ms": _percentile(lat_, 0.99), "max_ms": max(lat_), "calls": len(lat_),
                        "applications_evaluated_per_call": st_[2] / max(1, st_[0]), "applications_from_cache_per_call": st_[3] / max(1, st_[0])}

            warm_n = 5
            protos = set(args.fifo_protocols.split(","))
            rolled_all = [np.roll(fq, -i) for i in range(args.filter_calls + warm_n)]
            cold, st_cold = protocol([(q, N.ptr(q), n_q) for q in rolled_all])
            failed_cold = int(ffailed.value)
            span = min(256, n_q - 1)
            pq = N.ptr(fq)
            heads = [n_q - span + (i % span) for i in range(args.filter_calls + warm_n)]  # driver index j -> chain of j + 1 applications
            warm, st_warm = protocol([(fq, pq, j + 1) for j in heads]) if "warm" in protos else (cold, st_cold)
            retry, st_retry = protocol([(fq, pq, n_q)] * (min(args.filter_calls, 200) + warm_n)) if "retry" in protos else (cold, st_cold)
            ff = {"chain": f"{n_q - 1} earlier drivers + 1, tightly-pack, {args.nodes} nodes, host entry point incl. H2D/D2H",
                  "p50_ms": _percentile(cold[warm_n:], 0.5), "p99_ms": _percentile(cold[warm_n:], 0.99), "max_ms": max(cold[warm_n:]),
                  "calls": len(cold) - warm_n, "heads": "a different head (rotation of the queue) per call: every chain replays from the snapshot",
                  "decisions_per_s": n_q / (_percentile(cold[warm_n:], 0.5) * 1e-3), "chain_failed_at": failed_cold,
                  "cold_rotated_heads": summary(cold, st_cold, warm_n),
                  "warm_creation_order_heads": dict(summary(warm, st_warm, warm_n), heads=f"drivers {n_q - span} .. {n_q - 1} in creation order, cyclically; "
                                                    f"chains of {n_q - span + 1} .. {n_q} applications resumed from the previous chain's checkpoints"),
                  "warm_same_head": summary(retry, st_retry, warm_n)}
            # ---- the chain's roofline, over the time of the SHIPPED kernel: HIP events on the context's stream around the
            #      device work of whole cold Filters (the chain kernel and the small launches before / behind it), checkpoints
            #      off, rotated heads.  (Round 4 divided by the in-kernel cycle count of the instrumented variant, which runs
            #      longer than the Filter it is part of; that count now only splits the time into phases, in `phases`.)
            chain_ms = None
            try:
                ctx.set_option("chain_cache", 0)
                evs = []
                for i in range(12):
                    ctx.timer_begin(0)
                    chain_call(N.ptr(rolled_all[i]), n_q)
                    evs.append(ctx.timer_end())
                chain_ms = _median(evs[2:])
            except Exception:
                chain_ms = None
            finally:
                ctx.set_option("chain_cache", 1)
            cyc = ticks = 0
            if protos >= {"cold", "warm", "retry"}:  # (left out of the counter passes: their launches are all plain cold chains)
                ctx.set_option("chain_cache", 0)
                ctx.scan_stats(enable=True, reset=True)
                chain_call(N.ptr(rolled_all[1]), n_q)
                ctx.scan_stats(enable=False, reset=False)
                ctx.set_option("chain_cache", 1)
                cyc, ticks = ctx.last_fifo_clock
            pmc = load_profile("pmc_chain.json")
            instr_per_app = (pmc or {}).get("fit_fifo_solo_instructions_per_app")
            clock_ghz = (cyc / (ticks * 10.0)) if ticks else CLOCK_GHZ
            ISSUE = 4.3  # cycles between two instructions of a lone wavefront (tools/micro/probe_issue.hip)
            # the same three fractions as everywhere, per chain: ONE SIMD's issue slots — the other wavefronts of the workgroup
            # sleep at a barrier while wavefront 0 walks the chain
            cfr, cbound, cfrac = ({}, None, None)
            if chain_ms:
                cfr, cbound, cfrac = three_fractions(chain_ms * 1e-3, (pmc or {}).get("hbm_bytes_per_chain"),
                                                     (pmc or {}).get("l2_request_bytes_per_chain"),
                                                     instr_per_app * n_q if instr_per_app else None, simds=1)
            roofline["fifo_chain"] = {
                "kernel": "fit_fifo_solo_kernel<tightly-pack>", "kernel_ms": chain_ms,
                "kernel_ms_is": "HIP events on the context's stream around the device work of one cold Filter (gf_timer_begin / "
                                "gf_fit_batch / gf_timer_end), median of 10 rotated heads, checkpoints off: the shipped kernel",
                "kernel_ms_le_filter_p50": (chain_ms <= ff["p50_ms"]) if chain_ms else None,
                "bound": cbound, "frac": cfrac, "fractions": cfr,
                "achieved": cfr[cbound]["achieved"] if cbound else None, "peak": cfr[cbound]["peak"] if cbound else None,
                "unit": cfr[cbound]["unit"] if cbound else None,
                "traffic": (pmc or {}).get("hbm_bytes_per_chain"),
                "why": "one controlling wavefront (the chain is sequential in the applications, resource.go:224-262) on one SIMD: its "
                       "issue fraction is instructions / (1 SIMD x clock x kernel time); a LONE wavefront cannot issue faster than one "
                       "instruction per ~4.3 cycles, so `lone_wavefront_issue_frac` = instructions x 4.3 / (kernel time x clock) says "
                       "how much of the chain's time is issue at that rate (the rest: vector -> scalar -> branch hand-overs, LDS round "
                       "trips, taken branches — DESIGN.md 4.2; the instruction count is that of ALL the workgroup's wavefronts, "
                       "prologue and epilogue included, so the controlling wavefront's own share is lower)",
                "lone_wavefront_issue_frac": (instr_per_app * n_q * ISSUE / (chain_ms * 1e-3 * clock_ghz * 1e9))
                                             if (instr_per_app and chain_ms) else None,
                "filter_p50_ms": ff["p50_ms"], "filter_p99_ms": ff["p99_ms"],
                "filter_warm_p50_ms": ff["warm_creation_order_heads"]["p50_ms"], "filter_warm_p99_ms": ff["warm_creation_order_heads"]["p99_ms"],
                "filter_same_head_p50_ms": ff["warm_same_head"]["p50_ms"], "filter_same_head_p99_ms": ff["warm_same_head"]["p99_ms"],
                "applications_per_chain": n_q,
                "phases": {"instrumented_variant_shader_cycles_per_application": (cyc / n_q) if cyc else None,
                           "note": "in-kernel clock of the INSTRUMENTED variant (gf_scan_stats): slower than the shipped kernel, "
                                   "information only — never a divisor of frac"},
                "shader_clock_GHz": clock_ghz,
                "instructions_per_application": instr_per_app, "issue_cycles_per_instruction": ISSUE,
                "counters_from": (f"profiles/pmc_chain.json ({pmc.get('tag')}): rocprofv3 --pmc passes of `bench.py --no-extras "
                                  "--fifo-protocols cold` — a committed profile, not this run; the kernel time IS this run's") if pmc else None,
            }
            if not args.no_cpu_baseline:
                ff["cpu_baseline"] = cpu_chain_baseline(0, s.avail, s.sched, None, s.driver_order, s.exec_order, w.drv, w.exe, w.k,
                                                        w.flags, reps=20, with_eff_reps=5, maps_reps=3)
                ff["speedup_vs_cpu_p50"] = ff["cpu_baseline"]["p50_ms"] / ff["p50_ms"]
                ff["warm_speedup_vs_cpu_p50"] = ff["cpu_baseline"]["p50_ms"] / ff["warm_creation_order_heads"]["p50_ms"]
                if "reference_shaped_p50_ms" in ff["cpu_baseline"]:
                    ff["speedup_vs_reference_shaped_cpu_p50"] = ff["cpu_baseline"]["reference_shaped_p50_ms"] / ff["p50_ms"]
            out["fifo_filter"] = ff
        except Exception as e:
            out["fifo_filter"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- node-range sharding: ONE pending table evaluated by all ranks, each scanning its range of the
    #      priority order; three KB-sized exchanges per batch over RCCL/xGMI.  Strong scaling over nodes; reported next
    #      to the app-sharded headline so that the driver's 1/2/4/8-GPU runs measure both.
    # The headline above is measured; what follows is optional.  On N > 1 GPUs the optional leg talks over RCCL, and a
    # collective that never returns cannot be caught as an exception: a watchdog thread makes sure the one JSON line of
    # the contract still leaves rank 0 (and that no rank outlives the run) if that leg wedges.
    watchdog = None
    if world > 1 and not args.no_extras:
        import threading

        def _bail():
            if rank == 0:
                out["node_sharded"] = {"error": "timeout: the node-sharded leg did not finish within 240 s"}
                emit(out)
            os._exit(0)

        watchdog = threading.Timer(240.0 if rank == 0 else 250.0, _bail)
        watchdog.daemon = True
        watchdog.start()
    if not args.no_extras:
        try:
            from gangfit import sharded

            comm = sharded.TorchComm() if dist is not None else sharded.SingleComm()

            def time_sharded(c, wk, steps, warmup):
                eng = sharded.HipShardEngine(c, rank, world, dev)
                sb = sharded.ShardedBatch(eng, comm, TIGHT, gangfit.make_apps(wk.drv, wk.exe, wk.k, wk.flags))
                for _ in range(warmup):
                    sb.step()
                barrier()
                t0 = time.perf_counter()
                for _ in range(steps):
                    sb.step()
                barrier()
                wall_s = time.perf_counter() - t0
                if dist is not None:
                    t = torch.tensor([wall_s], dtype=torch.float64, device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    wall_s = float(t.item())
                # the three data-path exchanges of a batch, device time (events on the engine's stream), median of five batches
                try:
                    ex = [sb.step_timed() for _ in range(5)]
                    exchange_us = [_median([e[i] for e in ex]) for i in range(3)]
                except Exception:
                    exchange_us = None
                feas = None
                try:
                    feas = float(sb.fetch().results["has_capacity"].mean())
                except Exception:
                    pass
                return {"decisions_per_s": sb.n_apps * steps / wall_s, "ms_per_batch": wall_s / steps * 1e3,
                        "apps": sb.n_apps, "nodes": len(wk.snapshot.avail), "n_shards": world, "steps": steps,
                        "feasible_fraction": feas, "exchange_us": exchange_us,
                        "collectives_per_batch": "2 all-gather (16 B/app) + 1 all-reduce (4 B/executor)"}

            node_sharded = {"headline": time_sharded(ctx, base, max(10, min(args.steps, 200) // 4), 5)}
            # BASELINE config 4's size: 50 000 nodes x 10 000 apps (gang size = MinExecutorCount)
            w4 = wl.config(4)
            ctx4 = gangfit.Context(local_rank)
            ctx4.set_snapshot(w4.snapshot.avail, w4.snapshot.sched)
            ctx4.set_orders(w4.snapshot.driver_order, w4.snapshot.exec_order)
            node_sharded["config4_50k_nodes_x_10k_apps"] = time_sharded(ctx4, w4, 10, 2)
            # The same size on a nearly full cluster (usage U[0.93, 1]: the variant tests/test_gpu_fullsize.py checks against the
            # oracle): most gangs need the whole executor order — infeasible ones a full scan — so that every shard has work.  On the
            # nominal cluster tightly-pack drains the FRONT of the order: the first range does the scanning and sharding can only
            # cost.  This is the leg the contract line carries on N > 1 (config.node_sharded).
            w4c = wl.Workload("C4 congested", wl.make_snapshot(50000, 0x5EED0004, 0.93, 1.0), w4.drv, w4.exe, w4.k, w4.k_max, w4.flags)
            ctx4.set_snapshot(w4c.snapshot.avail, w4c.snapshot.sched)
            ctx4.set_orders(w4c.snapshot.driver_order, w4c.snapshot.exec_order)
            node_sharded["config4_congested_50k_nodes_x_10k_apps"] = time_sharded(ctx4, w4c, 10, 2)
            c4c = node_sharded["config4_congested_50k_nodes_x_10k_apps"]
            out["config"]["node_sharded"] = {
                "workload": "config 4 congested, 50k nodes x 10k apps, node-range shards", "n_shards": world,
                "ms_per_batch": c4c["ms_per_batch"], "exchange_us": c4c["exchange_us"], "feasible_fraction": c4c["feasible_fraction"],
                # ranks of the DATA-path collectives (two all-gathers + one all-reduce per batch), not of the timing barrier
                "rccl_ranks": (dist.get_world_size() if (dist is not None and backend == "nccl") else 0),
                "nominal_ms_per_batch": node_sharded["config4_50k_nodes_x_10k_apps"]["ms_per_batch"]}
            ctx4.set_snapshot(w4.snapshot.avail, w4.snapshot.sched)
            ctx4.set_orders(w4.snapshot.driver_order, w4.snapshot.exec_order)
            if world == 1:  # the unsharded kernel on the same table, for the cost of the four-step path itself
                a4, k4 = gangfit.with_offsets(gangfit.make_apps(w4.drv, w4.exe, w4.k, w4.flags))
                d_a4 = torch.from_numpy(a4.view(np.uint8).copy()).to(dev)
                d_r4 = torch.zeros(len(a4) * 16, dtype=torch.uint8, device=dev)
                d_e4 = torch.zeros(k4 + 1, dtype=torch.int32, device=dev)
                for _ in range(3):
                    ctx4.fit_batch_dev(IND, TIGHT, len(a4), d_a4.data_ptr(), d_r4.data_ptr(), d_e4.data_ptr(), k4, stream=stream)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    ctx4.fit_batch_dev(IND, TIGHT, len(a4), d_a4.data_ptr(), d_r4.data_ptr(), d_e4.data_ptr(), k4, stream=stream)
                torch.cuda.synchronize()
                node_sharded["config4_unsharded_one_gpu_decisions_per_s"] = len(a4) * 10 / (time.perf_counter() - t0)
                ctx4.set_snapshot(w4c.snapshot.avail, w4c.snapshot.sched)
                ctx4.set_orders(w4c.snapshot.driver_order, w4c.snapshot.exec_order)
                for _ in range(2):
                    ctx4.fit_batch_dev(IND, TIGHT, len(a4), d_a4.data_ptr(), d_r4.data_ptr(), d_e4.data_ptr(), k4, stream=stream)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    ctx4.fit_batch_dev(IND, TIGHT, len(a4), d_a4.data_ptr(), d_r4.data_ptr(), d_e4.data_ptr(), k4, stream=stream)
                torch.cuda.synchronize()
                node_sharded["config4_congested_unsharded_one_gpu_ms_per_batch"] = (time.perf_counter() - t0) / 5 * 1e3
                out["config"]["node_sharded"]["unsharded_one_gpu_ms_per_batch"] = node_sharded["config4_congested_unsharded_one_gpu_ms_per_batch"]
                ctx4.set_snapshot(w4.snapshot.avail, w4.snapshot.sched)
                ctx4.set_orders(w4.snapshot.driver_order, w4.snapshot.exec_order)
                # the dynamic-allocation tail of config 4: (max - min) single-executor first fits for 10 % of the apps
                # (rescheduleExecutor's loop, resource.go:658-662), as independent requests against the same snapshot
                extra = np.repeat(w4.exe[::10], np.maximum(w4.k_max[::10] - w4.k[::10], 0), axis=0)
                if len(extra):
                    ctx4.executor_fit(extra[:8])
                    t0 = time.perf_counter()
                    ctx4.executor_fit(extra)
                    node_sharded["config4_extra_executor_first_fits"] = {
                        "requests": int(len(extra)), "requests_per_s": len(extra) / (time.perf_counter() - t0),
                        "note": "host entry point incl. H2D/D2H"}
            ctx4.close()
            # ---- the same split INSIDE the library: one gf_ctx over several devices (gf_init with n_dev > 1, exchanges by peer
            #      access).  One GPU: N shards on this device (what the path itself costs).  N GPUs: rank 0 drives all N devices
            #      from one process while the other ranks wait at the barrier below; a subprocess, so that a fault in this
            #      optional leg cannot take the contract line down.
            if rank == 0:
                import subprocess

                grp = {}
                for cfgname in ("headline", "config4"):
                    devs = ",".join(str(i) for i in range(world)) if world > 1 else "0,0,0,0,0,0,0,0"
                    try:
                        env = dict(os.environ)
                        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                            env.pop(k, None)
                        p = subprocess.run([sys.executable, os.path.join(REPO, "tools", "group_bench.py"), "--devices", devs,
                                            "--config", cfgname, "--steps", "20"], capture_output=True, text=True, timeout=70, env=env)
                        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
                        grp[cfgname] = json.loads(line[-1]) if line else {"error": (p.stderr or p.stdout)[-400:]}
                    except Exception as e:
                        grp[cfgname] = {"error": f"{type(e).__name__}: {e}"}
                node_sharded["in_library_multi_device_context"] = grp
                # the verdict of the in-library node-range sharding in the contract line: how many shards served,
                # and whether each exchange's first sharded batch agreed with the first device's own answer
                g_h = grp.get("headline") or {}
                out["config"]["shard_count"] = (g_h.get("group") or {}).get("shard_count")
                out["config"]["shard_devices"] = len(set(g_h.get("devices") or []))
                out["config"]["group_selfcheck"] = {ex: bool((g_h.get(name) or {}).get("results_equal_one_device"))
                                                    for ex, name in (("peer_stores", "group"), ("rccl", "group_rccl")) if name in g_h}
            out["node_sharded"] = node_sharded
        except Exception as e:  # the headline number above must survive a failure of this optional leg
            out["node_sharded"] = {"error": f"{type(e).__name__}: {e}"}
    if watchdog is not None:
        watchdog.cancel()

    if rank == 0 and world == 1 and not args.no_extras:
        extras = {}
        out["extras"] = extras
        try:
            # One 1000-app launch occupies the chip for a few us, most of it the latency of a launch and of three dependent
            # misses, not work.  Four contexts (own snapshot copy, own stream: what four instance groups sharing one GPU
            # would be) interleave their launches; reported next to the headline, never instead of it.
            n_rep = 4
            reps = []
            for r in range(n_rep):
                c = gangfit.Context(local_rank)
                c.set_snapshot(s.avail, s.sched)
                c.set_orders(s.driver_order, s.exec_order)
                st = torch.cuda.Stream(device=dev)
                reps.append((c, st, torch.zeros(len(apps) * 16, dtype=torch.uint8, device=dev),
                             torch.zeros(total_k + 1, dtype=torch.int32, device=dev)))

            def rep_round():
                for c, st, r_res, r_exec in reps:
                    c.fit_batch_dev(IND, TIGHT, len(apps), d_apps.data_ptr(), r_res.data_ptr(), r_exec.data_ptr(), total_k,
                                    stream=st.cuda_stream)

            rounds = max(50, min(args.steps, 2000) // n_rep)
            for _ in range(20):
                rep_round()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(rounds):
                rep_round()
            torch.cuda.synchronize()
            wall_r = time.perf_counter() - t0
            same = all(bool(torch.equal(r_res, d_res)) for _, _, r_res, _ in reps)
            extras["four_replicas_one_gpu"] = {"decisions_per_s": n_rep * rounds * len(apps) / wall_r,
                                               "us_per_launch": wall_r / (n_rep * rounds) * 1e6,
                                               "launches": n_rep * rounds, "results_equal_headline": same}
            for c, _, _, _ in reps:
                c.close()
        except Exception as e:
            extras["four_replicas_one_gpu"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            # distribute-evenly on the same batch
            esteps = min(args.steps, 400)
            wall_e, kern_e, _, how_e = timed_graph(lambda: step(EVEN), esteps, max(2, args.warmup // 4), 3)
            extras["distribute_evenly"] = {"decisions_per_s": len(apps) * esteps / wall_e, "kernel_ms": kern_e, "submission": how_e}
            # congested cluster (usage ~U[0.95,1]): ~half of the gangs do not fit -> full scans + driver fallback
            wc = wl.headline(args.nodes, args.apps, congested=True)
            sc = wc.snapshot
            ctx.set_snapshot(sc.avail, sc.sched)
            ctx.set_orders(sc.driver_order, sc.exec_order)
            capps, ctotal = gangfit.with_offsets(gangfit.make_apps(wc.drv, wc.exe, wc.k, np.ones(len(wc.k), dtype=np.uint32)))
            d_capps = torch.from_numpy(capps.view(np.uint8).copy()).to(dev)
            d_cexec = torch.zeros(ctotal + 1, dtype=torch.int32, device=dev)

            def cstep():
                ctx.fit_batch_dev(IND, TIGHT, len(capps), d_capps.data_ptr(), d_res.data_ptr(), d_cexec.data_ptr(), ctotal,
                                  stream=stream)

            csteps = max(10, min(args.steps, 400) // 4)
            wall_c, kern_c, _, how_c = timed_graph(cstep, csteps, 3, 3)
            ctx.scan_stats(enable=True, reset=True)
            cstep()
            torch.cuda.synchronize()
            xv, dv = ctx.scan_stats(enable=False, reset=True)
            cb = wl.algorithmic_bytes(len(sc.exec_order), wc.k)
            cvis = xv * 24 + dv * 28 + len(capps) * 88 + 4 * int(wc.k.sum())
            res = d_res.cpu().numpy().view(gangfit._native.RESULT_DTYPE)
            clat, _ = fifo_latency(ctx, TIGHT, capps, 20, warm=2)
            extras["congested"] = {
                "workload": wc.name, "feasible_fraction": float(res["has_capacity"].mean()),
                "decisions_per_s": len(capps) * csteps / wall_c, "kernel_ms": kern_c, "submission": how_c,
                "information_only": {"visited_bytes_per_launch": cvis, "visited_GBps": cvis / (kern_c * 1e-3) / 1e9,
                                     "algorithmic_full_scan_GBps": cb / (kern_c * 1e-3) / 1e9,
                                     "note": "cache-served / unread bytes: never divided by a peak (no counter pass of this batch)"},
                "fifo_filter_p50_ms": _percentile(clat, 0.5), "fifo_filter_p99_ms": _percentile(clat, 0.99),
            }
            if not args.no_cpu_baseline:
                extras["congested"]["cpu_baseline"] = cpu_baseline_congested(wc)
                extras["congested"]["fifo_filter_cpu_baseline"] = cpu_chain_baseline(
                    0, sc.avail, None, None, sc.driver_order, sc.exec_order, wc.drv, wc.exe, wc.k, np.ones(len(wc.k), dtype=np.uint32),
                    reps=1)

            # ---- the rows of SURVEY.md 8f, each on the headline-sized cluster (nominal usage), each with its CPU chain
            def host_ms(f, n=20, warm=3):
                for _ in range(warm):
                    f()
                ts = []
                for _ in range(n):
                    t0 = time.perf_counter()
                    f()
                    ts.append((time.perf_counter() - t0) * 1e3)
                return _percentile(ts, 0.5), _percentile(ts, 0.99)

            happs = gangfit.make_apps(base.drv, base.exe, base.k, base.flags)
            zone3 = (wl.splitmix64(0xA3, len(s.avail), 9) % np.uint64(3)).astype(np.uint32)
            # three zones; the priority order is the reference's own: AZ-major (nodesorting.go:82-122 sorts by AZ priority first),
            # so every zone is a contiguous range of it.  The same zones scattered over the single-zone order (what rounds 1 and
            # 2 measured: no extender produces it) is kept as `zones_interleaved_*` for continuity.
            zorder = wl.reference_node_order(s.avail, zone3)
            ctx.set_snapshot(s.avail, s.sched)
            ctx.set_zones(zone3)
            SAZ, MF, SAZMF, AZA = (gangfit.GF_ALGO_SINGLE_AZ_TIGHTLY_PACK, gangfit.GF_ALGO_MINIMAL_FRAGMENTATION,
                                   gangfit.GF_ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION, gangfit.GF_ALGO_AZ_AWARE_TIGHTLY_PACK)
            for name, algo, n_fifo in (("single_az_tightly_pack", SAZ, 20), ("az_aware_tightly_pack", AZA, 10),
                                       ("minimal_fragmentation", MF, 10), ("single_az_minimal_fragmentation", SAZMF, 10)):
                zoned = algo != MF
                order = zorder if zoned else s.exec_order
                ctx.set_orders(order, order)
                p50, _ = host_ms(lambda: ctx.fit_batch(IND, algo, happs), n=10)
                flat, _ = fifo_latency(ctx, algo, happs, n_fifo, warm=1)
                extras[name] = {"zones": 3 if zoned else 1, "priority_order": "az-major (reference)" if zoned else "single zone (reference)",
                                "independent_decisions_per_s_host_entry": len(happs) / (p50 * 1e-3),
                                "fifo_filter_p50_ms": _percentile(flat, 0.5), "fifo_filter_p99_ms": _percentile(flat, 0.99)}
                if not args.no_cpu_baseline:
                    extras[name]["fifo_filter_cpu_baseline"] = cpu_chain_baseline(
                        int(algo), s.avail, s.sched, zone3 if zoned else None, order, order, base.drv, base.exe,
                        base.k, base.flags, reps=2)
                    extras[name]["speedup_vs_cpu_p50"] = extras[name]["fifo_filter_cpu_baseline"]["p50_ms"] / extras[name]["fifo_filter_p50_ms"]
                try:  # the lone feasibility call of the same packer (gf_fit_feasible: what UnschedulablePodMarker reads)
                    f50, _ = host_ms(lambda: ctx.fit_feasible(algo, happs), n=20)
                    extras[name]["feasibility_only_us_per_call_host_entry"] = f50 * 1e3
                except Exception as e:
                    extras[name]["feasibility_only_us_per_call_host_entry"] = f"{type(e).__name__}: {e}"
                # counter-backed rooflines of this packer's two kernels: the committed profile of ONE instantiation per
                # command (tools/profile_round.sh zoned -> profiles/pmc_zoned.json), three fractions over rocprofv3's average
                # dispatch duration of that command; this run's Filter p50 beside the chain's
                zp = (load_profile("pmc_zoned.json") or {})
                zk = zp.get("kernels") or {}
                keys = {"single_az_tightly_pack": ("zc_saz", "zb_saz"), "az_aware_tightly_pack": ("zc_aza", None),
                        "minimal_fragmentation": ("mc_mf", None), "single_az_minimal_fragmentation": ("mc_smf", "zb_smf")}[name]
                rf = {}
                for role, key in (("fifo_chain", keys[0]), ("independent_batch", keys[1])):
                    k = zk.get(key) if key else None
                    if not k:
                        continue
                    rf[role] = {"kernel": k.get("kernel"), "kernel_ms": (k.get("avg_dispatch_ns") or 0) * 1e-6, "bound": k.get("bound"),
                                "frac": k.get("frac"), "fractions": k.get("fractions"), "traffic": k.get("hbm_bytes"),
                                "wait_fraction": k.get("wait_fraction"),
                                "lds_bank_conflict_per_active_lds_cycle": k.get("lds_bank_conflict_per_active_lds_cycle"),
                                "simds": k.get("simds"), "counters_from": f"profiles/pmc_zoned.json ({k.get('tag') or zp.get('tag')}), entry {key}"}
                    if role == "fifo_chain":
                        rf[role]["this_run_filter_p50_ms"] = extras[name]["fifo_filter_p50_ms"]
                if rf:
                    extras[name]["roofline"] = rf
                if zoned:
                    ctx.set_orders(s.driver_order, s.exec_order)
                    flat, _ = fifo_latency(ctx, algo, happs, 4, warm=1)
                    extras[name]["zones_interleaved_fifo_filter_p50_ms"] = _percentile(flat, 0.5)
            ctx.set_orders(s.driver_order, s.exec_order)
            exe_reqs = np.ascontiguousarray(base.exe)
            p50, _ = host_ms(lambda: ctx.executor_fit(exe_reqs))
            m50, _ = host_ms(lambda: ctx.executor_fit(exe_reqs, minimal_fragmentation=True))
            extras["executor_fit"] = {"requests": len(exe_reqs), "first_fit_requests_per_s": len(exe_reqs) / (p50 * 1e-3),
                                      "minimal_fragmentation_requests_per_s": len(exe_reqs) / (m50 * 1e-3),
                                      "note": "host entry point incl. H2D/D2H"}
            f50, f99 = host_ms(lambda: ctx.find_nodes(base.exe[:200], base.k[:200], chained=True, want_adds=False), n=10)
            extras["find_nodes"] = {"requests": 200, "chained_p50_ms": f50, "chained_p99_ms": f99,
                                    "note": "failover.go:412-436 for 200 stale applications in a row, host entry point"}
            # snapshot construction on the device: reservation replay + metadata + priority orders + slot tables
            snap = {}
            for n_nodes, n_rr in ((10000, 2000), (100000, 20000)):
                rng = np.random.default_rng(n_nodes)
                shape = rng.integers(0, 4, size=n_nodes)
                alloc = np.stack([np.array([16, 32, 64, 96])[shape] * 1000, np.array([64, 128, 256, 384])[shape] * wl.GIB,
                                  np.zeros(n_nodes, dtype=np.int64)], axis=1).astype(np.int64)
                ks = rng.integers(2, 26, size=n_rr)
                rnode = rng.integers(0, n_nodes, size=int(ks.sum())).astype(np.uint32)
                rreq = np.stack([rng.choice([1000, 2000, 4000], size=len(rnode)), rng.choice([4, 8, 16], size=len(rnode)) * wl.GIB,
                                 np.zeros(len(rnode), dtype=np.int64)], axis=1).astype(np.int64)
                flags = np.full(n_nodes, 2 | 4, dtype=np.uint32)
                ranks = rng.permutation(n_nodes).astype(np.uint32)
                zone = rng.integers(0, 3, size=n_nodes).astype(np.uint32)
                p50, p99 = host_ms(lambda: ctx.build_snapshot(alloc, flags, ranks, res_node=rnode, res_req=rreq, zone=zone,
                                                              n_zones=3), n=100, warm=3)
                snap[f"{n_nodes}_nodes_{n_rr}_reservations"] = {"reservation_entries": int(len(rnode)), "p50_ms": p50, "p99_ms": p99,
                                                                "calls": 100}
            extras["snapshot_build"] = snap
            # ---- BASELINE config 5: 100 000 nodes, 20 000 ResourceReservations replayed, FIFO 999 + 1 (5 % skippable)
            w5 = wl.config(5)
            n5 = len(w5.snapshot.avail)
            rng = np.random.default_rng(5)
            ks = rng.integers(2, 26, size=20000)
            rnode = rng.integers(0, n5, size=int(ks.sum())).astype(np.uint32)
            rreq = np.stack([rng.choice([1000, 2000, 4000], size=len(rnode)), rng.choice([4, 8, 16], size=len(rnode)) * wl.GIB,
                             np.zeros(len(rnode), dtype=np.int64)], axis=1).astype(np.int64)
            flags5 = np.full(n5, 2 | 4, dtype=np.uint32)
            ranks5 = np.arange(n5, dtype=np.uint32)
            alloc5 = w5.snapshot.sched + 0  # allocatable of the synthetic cluster; usage comes from the replayed reservations
            q5 = gangfit.make_apps(w5.drv, w5.exe, w5.k, w5.flags)
            lat5, chain5, o5 = [], [], None
            n_calls5 = min(args.filter_calls, 300)
            # the cluster's static columns (allocatable, flags, name ranks) stay resident (gf_cluster_set), the reservation
            # entries travel as the columns the C ABI takes: a Filter moves the reservations and the queue, nothing else
            ctx.set_cluster(alloc5, flags5, ranks5)
            rcols5 = [np.ascontiguousarray(rreq[:, j]) for j in range(3)]
            for i in range(n_calls5 + 3):
                rolled = np.roll(q5, -i)
                t0 = time.perf_counter()
                if i == 0:  # once with the order lists (the CPU leg below needs them)
                    D5, X5 = ctx.build_snapshot_resident(res_node=rnode, res_cols=rcols5)
                else:       # steady state: nothing of size O(n_nodes) returns to the host
                    ctx.build_snapshot_resident(res_node=rnode, res_cols=rcols5, want_orders=False)
                t1 = time.perf_counter()
                o5 = ctx.fit_batch(FIFO, TIGHT, rolled)
                t2 = time.perf_counter()
                if i >= 3:
                    lat5.append((t2 - t0) * 1e3)
            # the same Filter with the usage sums resident too (gf_usage_apply): what changes between two Filters is a
            # handful of reservations — here one application's K + 1 entries leave and another one's arrive per call
            ctx.usage_reset()
            ctx.usage_apply(rnode, res_cols=rcols5, sign=+1)
            latu = []
            starts5 = np.concatenate([[0], np.cumsum(ks)])
            for i in range(min(n_calls5, 100) + 3):
                rolled = np.roll(q5, -i)
                j = i % len(ks)
                sl = slice(int(starts5[j]), int(starts5[j + 1]))
                dn, dc = rnode[sl], [c[sl] for c in rcols5]
                t0 = time.perf_counter()
                ctx.usage_apply(dn, res_cols=dc, sign=-1)   # this application's reservations went away ...
                ctx.usage_apply(dn, res_cols=dc, sign=+1)   # ... and (the same entries, as another application's) arrived
                ctx.build_snapshot_resident(resident_usage=True, want_orders=False)
                ctx.fit_batch(FIFO, TIGHT, rolled)
                if i >= 3:
                    latu.append((time.perf_counter() - t0) * 1e3)
            # the same Filter with a device synchronise behind every phase (so that a phase's kernels are charged to it), and the
            # chain by itself on the snapshot the last build left — over the SAME heads as the loop above: the chain's time
            # depends on the head (3.6 .. 6.1 ms over the rotations of this queue, two clusters), so medians over different
            # sets of heads are not comparable (round 3 compared 100 heads with 60 and read the difference as a regression)
            n_u = min(n_calls5, 100) + 3
            ph5 = {"usage_apply_x2": [], "snapshot_build_resident": [], "chain_first_on_fresh_epoch": []}
            for i in range(n_u):
                rolled = np.roll(q5, -i)
                j = i % len(ks)
                sl = slice(int(starts5[j]), int(starts5[j + 1]))
                dn, dc = rnode[sl], [c[sl] for c in rcols5]
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ctx.usage_apply(dn, res_cols=dc, sign=-1)
                ctx.usage_apply(dn, res_cols=dc, sign=+1)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                ctx.build_snapshot_resident(resident_usage=True, want_orders=False)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                ctx.fit_batch(FIFO, TIGHT, rolled)
                t3 = time.perf_counter()
                if i >= 3:
                    ph5["usage_apply_x2"].append((t1 - t0) * 1e3)
                    ph5["snapshot_build_resident"].append((t2 - t1) * 1e3)
                    ph5["chain_first_on_fresh_epoch"].append((t3 - t2) * 1e3)
            for i in range(n_u):
                rolled = np.roll(q5, -i)
                t1 = time.perf_counter()
                ctx.fit_batch(FIFO, TIGHT, rolled)
                if i >= 3:
                    chain5.append((time.perf_counter() - t1) * 1e3)
            ctx.set_option("chain_cache", 0)  # ... and without the checkpoints a chain dumps for the next Filter
            nock5 = []
            for i in range(n_u):
                rolled = np.roll(q5, -i)
                t1 = time.perf_counter()
                ctx.fit_batch(FIFO, TIGHT, rolled)
                if i >= 3:
                    nock5.append((time.perf_counter() - t1) * 1e3)
            ctx.set_option("chain_cache", 1)
            c5 = {"nodes": n5, "reservation_entries": int(len(rnode)), "earlier_drivers": len(q5) - 1, "calls": len(lat5),
                  "filter_p50_ms": _percentile(lat5, 0.5), "filter_p99_ms": _percentile(lat5, 0.99),
                  "chain_only_p50_ms": _percentile(chain5, 0.5), "chain_only_p99_ms": _percentile(chain5, 0.99),
                  "chain_only_quartiles_ms": [_percentile(chain5, q) for q in (0.0, 0.25, 0.5, 0.75, 1.0)],
                  "chain_only_no_checkpoints_p50_ms": _percentile(nock5, 0.5),
                  "chain_only_heads": f"the {len(chain5)} heads of the resident-usage loop (same rotations: the chain's time depends on the head)",
                  "filter_resident_usage_p50_ms": _percentile(latu, 0.5), "filter_resident_usage_p99_ms": _percentile(latu, 0.99),
                  "filter_resident_usage_phases_p50_ms": {k: _percentile(v, 0.5) for k, v in ph5.items()},
                  "filter_resident_usage_phases_p99_ms": {k: _percentile(v, 0.99) for k, v in ph5.items()},
                  "filter_resident_usage": "two gf_usage_apply calls (one application's entries out, one's in) + "
                                           "gf_snapshot_build_resident(GF_RESIDENT_USAGE) + the chain: no reservation list travels",
                  "filter": "gf_snapshot_build_resident (reservation replay + metadata + sort + slot tables on the device; the cluster's "
                            "static columns resident: gf_cluster_set) + the FIFO chain, host entry points incl. H2D/D2H",
                  "chain_failed_at": o5.failed_at}
            if not args.no_cpu_baseline:
                a5, _ = ctx.snapshot()
                c5["cpu_baseline_chain"] = cpu_chain_baseline(0, a5, None, None, D5, X5, w5.drv, w5.exe, w5.k, w5.flags, reps=3)
            extras["config5_100k_nodes_fifo"] = c5
            # BASELINE config 3: 10 000 nodes x 10 000 pending apps, both plain packers, device resident
            extras["config3_10k_nodes_x_10k_apps"] = run_config3(ctx, torch, dev, stream, timed_graph)
        except Exception as e:  # keep what was measured; the headline line must still be printed
            import traceback

            extras["error"] = f"{type(e).__name__}: {e}"
            extras["error_where"] = traceback.format_exc().strip().splitlines()[-3:]

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl.headline(args.nodes, args.apps, seed=0x5EED0010))
        fcb = (out.get("fifo_filter") or {}).get("cpu_baseline")
        if fcb:  # the latency half of the metric, next to its GPU figures in roofline.fifo_chain
            out["cpu_baseline"]["fifo_chain"] = {
                "literal_p50_ms": fcb.get("p50_ms"), "with_efficiency_maps_p50_ms": fcb.get("with_efficiency_maps_p50_ms"),
                "reference_shaped_p50_ms": fcb.get("reference_shaped_p50_ms"), "cores": 1, "kind": "port",
                "sample": fcb.get("sample"),
                "note": "the reference replays every earlier driver on every Filter (resource.go:309-328): the CPU figure is the "
                        "same for cold and warm heads",
                "gpu_cold_p50_ms": out["fifo_filter"].get("p50_ms"),
                "gpu_warm_p50_ms": out["fifo_filter"].get("warm_creation_order_heads", {}).get("p50_ms"),
                "speedup_cold_p50": out["fifo_filter"].get("speedup_vs_cpu_p50"),
                "speedup_warm_p50": out["fifo_filter"].get("warm_speedup_vs_cpu_p50")}
        if not args.no_extras:
            try:
                out["cpu_baseline_variants"] = cpu_baseline_variants(args.nodes, args.apps)
            except Exception as e:
                out["cpu_baseline_variants"] = {"error": f"{type(e).__name__}: {e}"}

    if graph is not None:
        ctx.graph_destroy(graph)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(out)


if __name__ == "__main__":
    main()
