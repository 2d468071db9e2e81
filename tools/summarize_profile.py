#!/usr/bin/env python
"""Turn the rocprofv3 CSVs of tools/profile_round.sh (gpurun_out/prof_<tag>/) into the tracked summaries under profiles/:
  profiles/<tag>_kernel_stats*.csv  — rocprofv3 --kernel-trace --stats summaries (verbatim; one per profiled command)
  profiles/<tag>_pmc.csv            — per-kernel means of every PMC pass (one counter set per pass)
  profiles/<tag>_summary.md         — all of it, human readable
  profiles/pmc_headline.json        — per STEP counters of the two kernels behind bench.py's headline: one dispatch of
                                      fit_independent_kernel = one step; one dispatch of fit_worker_kernel = K tickets = K steps
  profiles/pmc_chain.json           — per chain counters of fit_fifo_solo_kernel (every launch replays the headline chain)
  profiles/pmc_config3.json         — per launch counters of BASELINE config 3, per packer
bench.py reads the three JSON files for its roofline objects (`fractions`, `traffic`).

Units and corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950
FETCH_SIZE counts a 128-byte request as 64 bytes for wide coalesced reads, so read bytes = 2 x FETCH_SIZE x 1024 (the upper
estimate; the raw figure is recorded next to it); hbm_bytes = 2 x fetch + write.  L2 request bytes = (TCC_HIT_sum +
TCC_MISS_sum) x 128.  instructions = SQ_INSTS_VALU + SQ_INSTS_SALU + SQ_INSTS_LDS + SQ_INSTS_SMEM (all wavefronts).
"""
import collections
import csv
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = ("fetch", "write", "l2", "sq", "busy", "sq2")
INST = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM")


def _rows(src, d):
    f = os.path.join(src, d, "pmc_counter_collection.csv")
    return list(csv.DictReader(open(f))) if os.path.exists(f) else []


def _bench_json(log):
    """The one JSON line bench.py printed under rocprofv3 (its stdout is in the pass's .log)."""
    if not os.path.exists(log):
        return None
    for line in reversed(open(log, errors="replace").read().splitlines()):
        if line.startswith("{"):
            try:
                return json.loads(line)
            except Exception:
                return None
    return None


def _group_means(src, pre):
    """{kernel: {counter: [values per dispatch]}} over the four counter passes of one profiled command."""
    per = collections.OrderedDict()
    for p in PASSES:
        for r in _rows(src, f"{pre}_{p}"):
            per.setdefault(r["Kernel_Name"], collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return per


def _derive(mean, div=1.0):
    """Per-step figures from a kernel's mean counters per dispatch (div = steps per dispatch)."""
    out = {k: v / div for k, v in mean.items()}
    rd, wr = mean.get("FETCH_SIZE"), mean.get("WRITE_SIZE")
    if rd is not None and wr is not None:
        out["fetch_bytes_raw"] = rd * 1024 / div
        out["write_bytes"] = wr * 1024 / div
        out["hbm_bytes"] = (2 * rd * 1024 + wr * 1024) / div
    if "TCC_HIT_sum" in mean and "TCC_MISS_sum" in mean:
        out["l2_request_bytes"] = 128 * (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"]) / div
    if any(c in mean for c in INST):
        out["instructions"] = sum(mean.get(c, 0.0) for c in INST) / div
    if mean.get("SQ_WAVE_CYCLES"):
        out["wait_fraction"] = mean.get("SQ_WAIT_ANY", 0.0) / mean["SQ_WAVE_CYCLES"]
    # rocprofv3's derived metrics (percent of the kernel's GPU time): NOT divided by the steps of a dispatch
    if mean.get("VALUBusy") is not None:
        out["valu_busy"] = mean["VALUBusy"] / 100.0
        out["VALUBusy"] = mean["VALUBusy"]
    if mean.get("SALUBusy") is not None:
        out["salu_busy"] = mean["SALUBusy"] / 100.0
        out["SALUBusy"] = mean["SALUBusy"]
    return out


def _med(v):
    v = sorted(v)
    n = len(v)
    return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])


def _durations(trace_csv, name):
    """End - Start of every dispatch of kernels whose name starts with `name`, in launch order (rocprofv3 --kernel-trace CSV)."""
    if not os.path.exists(trace_csv):
        return []
    out = []
    for r in csv.DictReader(open(trace_csv)):
        if r.get("Kernel_Name", "").startswith(name):
            out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return [float(d) for _, d in sorted(out)]


def _stats_table(path, want):
    md = "| kernel | calls | avg ns | min ns | max ns | % |\n|---|---|---|---|---|---|\n"
    for r in csv.DictReader(open(path)):
        if want is None or any(w in r["Name"] for w in want):
            md += (f"| {r['Name']} | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | "
                   f"{float(r['Percentage']):.3f} |\n")
    return md


def _avg_ns(path, name):
    if not os.path.exists(path):
        return None, None
    for r in csv.DictReader(open(path)):
        if r["Name"].startswith(name):
            return float(r["AverageNs"]), int(r["Calls"])
    return None, None


def main():
    # summarize_profile.py <tag> [<src dir> [<dst dir>]]: tools/profile_round.sh runs it ON THE BOX with dst = <src>/summary and
    # deletes the raw rocprofv3 CSVs afterwards (a K = 2 000 counter pass has 10^5 rows; gpurun copies back 64 MiB at most);
    # the builder then copies gpurun_out/prof_<tag>/summary/* into profiles/.
    tag = sys.argv[1] if len(sys.argv) > 1 else "r5"
    src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(REPO, "gpurun_out", f"prof_{tag}")
    dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(REPO, "profiles")
    os.makedirs(dst, exist_ok=True)
    md = (f"# rocprofv3 summary — {tag}\n\nRecipe: `tools/profile_round.sh {tag}` on one MI355X (gfx950): per profiled command one "
          "`rocprofv3 --kernel-trace --stats` run and one `--pmc` run per counter set (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum "
          "TCC_MISS_sum | the SQ set), never combined with other trace domains.\n\n")
    pmc_csv = [["command", "kernel", "counter", "dispatches", "mean", "min", "max"]]

    def dump_group(pre, per):
        for k, cs in per.items():
            for c, v in cs.items():
                pmc_csv.append([pre, k, c, len(v), f"{sum(v) / len(v):.3f}", f"{min(v):.3f}", f"{max(v):.3f}"])

    # ---- the headline, both regimes, at the driver's command (K = 20) and at bench.py's default (K = 2 000)
    runs = collections.OrderedDict()
    for pre, cmd in (("hl20", "bench.py --headline-only --steps 20 --warmup 5   (the driver's `--steps 20 --warmup 5`, headline leg only)"),
                     ("hl", "bench.py --headline-only   (bench.py's defaults: --steps 2000 --warmup 50)")):
        per = _group_means(src, pre)
        hstats = os.path.join(src, f"{pre}_stats", "stats_kernel_stats.csv")
        if not per and not os.path.exists(hstats):
            continue
        dump_group(pre, per)
        bj = _bench_json(os.path.join(src, f"{pre}_stats.log")) or {}
        steps = int(bj.get("steps") or (20 if pre == "hl20" else 2000))
        hj = {"command": cmd, "steps": steps}
        if "fit_independent_kernel" in per:
            mean = {c: sum(v) / len(v) for c, v in per["fit_independent_kernel"].items()}
            hj["launch_path"] = dict(_derive(mean, 1.0), kernel="fit_independent_kernel<tightly-pack>", steps_per_dispatch=1,
                                     dispatches=len(next(iter(per["fit_independent_kernel"].values()))), statistic="mean over the dispatches")
        if "fit_worker_kernel" in per:
            # every dispatch serves one window's K tickets; the MEDIAN over the dispatches (warm-up windows and the first, cold
            # one are among them: the mean is theirs)
            med = {c: _med(v) for c, v in per["fit_worker_kernel"].items()}
            hj["worker"] = dict(_derive(med, float(steps)), kernel="fit_worker_kernel<tightly-pack>", steps_per_dispatch=steps,
                                dispatches=len(next(iter(per["fit_worker_kernel"].values()))), statistic="median over the dispatches",
                                note="one dispatch serves the K tickets of a window (launch and departure included); its "
                                     "instruction and L2 counts include the polling of idle workgroups")
        if os.path.exists(hstats):
            shutil.copy(hstats, os.path.join(dst, f"{tag}_kernel_stats_headline_steps{steps}.csv"))
            trace = os.path.join(src, f"{pre}_stats", "stats_kernel_trace.csv")
            a, n = _avg_ns(hstats, "fit_independent_kernel")
            d = _durations(trace, "fit_independent_kernel")
            if a and "launch_path" in hj:
                hj["launch_path"]["rocprof_avg_dispatch_ns"], hj["launch_path"]["rocprof_dispatches"] = a, n
                if d:
                    hj["launch_path"]["rocprof_median_dispatch_ns"], hj["launch_path"]["rocprof_min_dispatch_ns"] = _med(d), min(d)
                lp = hj["launch_path"]
                t = (lp.get("rocprof_median_dispatch_ns") or a) * 1e-9
                lp["fractions_over_own_duration"] = {k: v for k, v in (
                    ("hbm", lp["hbm_bytes"] / t / 8e12 if lp.get("hbm_bytes") is not None else None),
                    ("l2", lp["l2_request_bytes"] / t / 34.5e12 if lp.get("l2_request_bytes") is not None else None),
                    ("issue", lp["instructions"] / t / (1024 * 2.4e9) if lp.get("instructions") is not None else None),
                    ("valu", lp.get("valu_busy"))) if v is not None}
            a, n = _avg_ns(hstats, "fit_worker_kernel")
            d = _durations(trace, "fit_worker_kernel")
            # the dispatches that served a whole K-ticket window: within a quarter of the traced run's own kernel time per window
            # (HIP events, the line's roofline.kernel_ms x K); the others are the short counter window and relaunches
            ref_ns = ((bj.get("roofline") or {}).get("kernel_ms") or 0) * 1e6 * steps
            if d and ref_ns and (bj.get("config") or {}).get("regime", "").startswith("streamed"):
                full = [x for x in d if 0.75 * ref_ns <= x <= 1.25 * ref_ns]
                if len(full) >= 3:
                    hj.setdefault("worker", {})["rocprof_dispatches_all_ns"] = d
                    d = full
            if a and "worker" in hj:
                wk = hj["worker"]
                wk["rocprof_avg_dispatch_ns"], wk["rocprof_dispatches"] = a, n
                if d:
                    wk["rocprof_dispatch_ns"] = d
                    wk["rocprof_median_dispatch_ns"], wk["rocprof_min_dispatch_ns"], wk["rocprof_max_dispatch_ns"] = _med(d), min(d), max(d)
                ref = _med(d) if d else a
                wk["rocprof_ns_per_ticket"] = ref / steps
                wk["rocprof_ns_per_ticket_is"] = "median dispatch duration / K" if d else "average dispatch duration / K"
                if d:
                    wk["rocprof_min_ns_per_ticket"] = min(d) / steps
                # the fractions a reader gets from THIS file alone: the profile's counters over the profile's own duration
                t = wk["rocprof_ns_per_ticket"] * 1e-9
                own = {}
                if wk.get("hbm_bytes") is not None:
                    own["hbm"] = wk["hbm_bytes"] / t / 8e12
                if wk.get("l2_request_bytes") is not None:
                    own["l2"] = wk["l2_request_bytes"] / t / 34.5e12
                if wk.get("instructions") is not None:
                    own["issue"] = wk["instructions"] / t / (1024 * 2.4e9)
                if wk.get("valu_busy") is not None:
                    own["valu"] = wk["valu_busy"]
                wk["fractions_over_own_duration"] = own
                # the traced run's own line: the kernel cannot take longer per ticket than a step of the window it served
                if bj.get("ms_per_step") and (bj.get("config") or {}).get("regime", "").startswith("streamed"):
                    wk["traced_run_ms_per_step"] = bj["ms_per_step"]
                    wk["traced_run_kernel_ms"] = (bj.get("roofline") or {}).get("kernel_ms")
                    wk["ns_per_ticket_le_ms_per_step"] = bool(wk["rocprof_ns_per_ticket"] <= bj["ms_per_step"] * 1e6 * 1.02)
                    assert wk["ns_per_ticket_le_ms_per_step"], (
                        f"{pre}: rocprofv3's median fit_worker_kernel dispatch / K = {wk['rocprof_ns_per_ticket']:.0f} ns per ticket exceeds the "
                        f"traced run's own ms_per_step = {bj['ms_per_step'] * 1e6:.0f} ns: the profile does not describe the line's kernel")
            md += (f"## the headline, both regimes (`{cmd}`)\n\nEvery `fit_independent_kernel` dispatch is "
                   f"one headline batch (launch path); every `fit_worker_kernel` dispatch serves one window's {steps} tickets.\n\n" +
                   _stats_table(hstats, ("fit_independent", "fit_worker", "empty_kernel")) + "\n")
            if d:
                md += (f"`fit_worker_kernel` dispatches (ns, launch order): {', '.join(str(int(x)) for x in d)} — median {_med(d):.0f}, "
                       f"min {min(d):.0f} = {_med(d) / steps:.0f} / {min(d) / steps:.0f} ns per ticket\n\n")
            if bj:
                rf = bj.get("roofline") or {}
                md += (f"bench line of that (traced) run: value {bj.get('value', 0) / 1e6:.1f} M decisions/s, "
                       f"{bj.get('ms_per_step', 0) * 1e3:.2f} us per step, roofline.kernel {rf.get('kernel')} "
                       f"kernel_ms {rf.get('kernel_ms')}\n\n")
        runs[f"steps{steps}"] = hj
    if runs:
        first = next(iter(runs.values()))
        hj = {"tag": tag,
              "note": "per STEP (one 1 000-application batch); hbm_bytes = 2 x FETCH_SIZE KiB (gfx950 correction) + WRITE_SIZE KiB; "
                      "l2_request_bytes = (TCC_HIT_sum + TCC_MISS_sum) x 128; instructions = SQ_INSTS_VALU + SALU + LDS + SMEM.  `runs` "
                      "holds one entry per profiled command (bench.py picks the one with its own --steps); the top-level launch_path / "
                      "worker repeat the driver's command (steps20).",
              "command": first["command"], "runs": runs}
        for k in ("launch_path", "worker"):
            if k in first:
                hj[k] = first[k]
        json.dump(hj, open(os.path.join(dst, "pmc_headline.json"), "w"), indent=1)
        md += "```json\n" + json.dumps(hj, indent=1) + "\n```\n\n"

    # ---- the plain FIFO chain alone
    per = _group_means(src, "chain")
    cstats = os.path.join(src, "chain_stats", "stats_kernel_stats.csv")
    solo = next((k for k in per if k.startswith("fit_fifo_solo_kernel")), None)
    if solo:
        dump_group("chain", per)
        mean = {c: sum(v) / len(v) for c, v in per[solo].items()}
        bj = _bench_json(os.path.join(src, "chain_sq.log")) or {}
        n_apps = ((bj.get("roofline") or {}).get("fifo_chain") or {}).get("applications_per_chain") or 1000
        d = _derive(mean, 1.0)
        cj = {"tag": tag, "kernel": "fit_fifo_solo_kernel", "launches": len(next(iter(per[solo].values()))),
              "applications_per_chain": n_apps, "per_launch": mean,
              "fit_fifo_solo_instructions_per_app": d.get("instructions", 0.0) / n_apps,
              "hbm_bytes_per_chain": d.get("hbm_bytes"), "l2_request_bytes_per_chain": d.get("l2_request_bytes"),
              "wait_fraction": d.get("wait_fraction"),
              "note": "VALU + SALU + LDS + SMEM instructions of ALL wavefronts of the workgroup per application (the helpers only run "
                      "the prologue, the checkpoint dumps and the epilogue); every launch is a full replay of the headline chain "
                      "(bench.py --fifo-protocols cold)"}
        if os.path.exists(cstats):
            shutil.copy(cstats, os.path.join(dst, f"{tag}_kernel_stats_chain.csv"))
            a, n = _avg_ns(cstats, "fit_fifo_solo_kernel")
            if a:
                cj["kernel_avg_ns"], cj["kernel_calls"] = a, n
        json.dump(cj, open(os.path.join(dst, "pmc_chain.json"), "w"), indent=1)
        md += ("## the plain FIFO chain alone (`--no-extras --fifo-protocols cold`: every launch replays the headline chain)\n\n"
               "```json\n" + json.dumps(cj, indent=1) + "\n```\n\n")

    # ---- config 3: the two packers are two instantiations of fit_independent_kernel (Kernel_Id in launch order)
    c3 = {"tag": tag, "note": "per launch; hbm_bytes = 2 x FETCH_SIZE KiB (gfx950 correction) + WRITE_SIZE KiB; L2 requests are "
                              "128-byte lines"}
    names = ["tightly_pack", "distribute_evenly"]
    for p in PASSES:
        rows = [r for r in _rows(src, f"c3_{p}") if r["Kernel_Name"].startswith("fit_independent_kernel")]
        ids = []
        for r in rows:
            if r["Kernel_Id"] not in ids:
                ids.append(r["Kernel_Id"])
        for i, kid in enumerate(ids[:2]):
            for c in sorted({r["Counter_Name"] for r in rows}):
                v = [float(r["Counter_Value"]) for r in rows if r["Kernel_Id"] == kid and r["Counter_Name"] == c]
                if v:
                    c3.setdefault(names[i], {})[c] = sum(v) / len(v)
                    c3[names[i]]["launches"] = len(v)
                    pmc_csv.append(["config3:" + names[i], "fit_independent_kernel", c, len(v), f"{sum(v) / len(v):.3f}",
                                    f"{min(v):.3f}", f"{max(v):.3f}"])
    for nme in names:
        t = c3.get(nme)
        if t:
            t.update({k: v for k, v in _derive(t, 1.0).items() if k in ("fetch_bytes_raw", "write_bytes", "hbm_bytes", "l2_request_bytes",
                                                                       "instructions", "wait_fraction", "valu_busy", "salu_busy")})
    kst = os.path.join(src, "c3_stats", "stats_kernel_stats.csv")
    if os.path.exists(kst):
        shutil.copy(kst, os.path.join(dst, f"{tag}_kernel_stats_config3.csv"))
    if any(n in c3 for n in names):
        json.dump(c3, open(os.path.join(dst, "pmc_config3.json"), "w"), indent=1)
        md += ("## BASELINE config 3 alone (`bench.py --config3-only`: 10 000 nodes x 10 000 apps, both packers)\n\n" +
               (_stats_table(kst, ("fit_independent",)) + "\n" if os.path.exists(kst) else "") +
               "```json\n" + json.dumps(c3, indent=1) + "\n```\n\n")

    # ---- the zone-aware / minimal-fragmentation kernels, one instantiation per profiled command (tools/profile_cmd.py)
    ZONED = (("zb_saz", "fit_zoned_fused_kernel", "single-az-tightly-pack", "independent batch, one launch", 1024),
             ("zb_smf", "fit_zoned_fused_kernel", "single-az-minimal-fragmentation", "independent batch, one launch", 1024),
             ("zb_mf", "fit_independent_kernel", "minimal-fragmentation", "independent batch, a workgroup per application (round 6)", 1024),
             ("zc_saz", "fit_fifo_zoned_lds_kernel", "single-az-tightly-pack", "cold FIFO chain", 4),
             ("zc_aza", "fit_fifo_zoned_lds_kernel", "az-aware-tightly-pack", "cold FIFO chain", 4),
             ("mc_mf", "fit_fifo_minfrag_lds_kernel", "minimal-fragmentation", "cold FIFO chain (8-wavefront instantiation)", 4),
             ("mc_smf", "fit_fifo_minfrag_lds_kernel", "single-az-minimal-fragmentation", "cold FIFO chain (16-wavefront instantiation)", 4))
    zj = {"tag": tag, "note": "per launch of ONE instantiation (tools/profile_cmd.py: 10 000 nodes x 1 000 applications, 3 zones, AZ-major order); "
                              "hbm_bytes = 2 x FETCH_SIZE KiB (gfx950 correction) + WRITE_SIZE KiB; l2_request_bytes = (TCC_HIT_sum + TCC_MISS_sum) x 128; "
                              "instructions = SQ_INSTS_VALU + SALU + LDS + SMEM of all wavefronts; fractions over rocprofv3's average dispatch "
                              "duration of the same command: hbm / 8 TB/s, l2 / 34.5 TB/s, issue / (simds x 2.4 GHz) with simds = 1 024 for a "
                              "grid-wide kernel and 4 (one compute unit) for a one-workgroup chain kernel",
          "kernels": {}}
    # a round that profiles only some instantiations (ZONED_SPECS of tools/profile_round.sh) keeps the others' entries, each under the
    # tag of the visit that produced it
    old_zoned = os.path.join(REPO, "profiles", "pmc_zoned.json")
    if os.path.exists(old_zoned):
        try:
            oz = json.load(open(old_zoned))
            for pre0, e0 in (oz.get("kernels") or {}).items():
                e0.setdefault("tag", oz.get("tag"))
                zj["kernels"][pre0] = e0
        except Exception:
            pass
    fresh = 0
    zmd = ""
    for pre, kname, algo, what, simds in ZONED:
        per = _group_means(src, pre)
        for r in _rows(src, f"{pre}_lds"):
            per.setdefault(r["Kernel_Name"], collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        k = next((x for x in per if x.startswith(kname)), None)
        st = os.path.join(src, f"{pre}_stats", "stats_kernel_stats.csv")
        if k is None:
            continue
        dump_group(pre, {k: per[k]})
        mean = {c: sum(v) / len(v) for c, v in per[k].items()}
        d = _derive(mean, 1.0)
        a, n = _avg_ns(st, kname)
        e = {"kernel": kname, "algo": algo, "what": what, "launches": len(next(iter(per[k].values()))), "avg_dispatch_ns": a, "dispatches_in_stats": n,
             "hbm_bytes": d.get("hbm_bytes"), "l2_request_bytes": d.get("l2_request_bytes"), "instructions": d.get("instructions"),
             "wait_fraction": d.get("wait_fraction"), "simds": simds,
             "lds_bank_conflict_per_active_lds_cycle": (mean["SQ_LDS_BANK_CONFLICT"] / mean["SQ_ACTIVE_INST_LDS"]) if mean.get("SQ_ACTIVE_INST_LDS") else None,
             "counters": mean}
        if a:
            t = a * 1e-9
            fr = {}
            if e["hbm_bytes"] is not None:
                fr["hbm"] = e["hbm_bytes"] / t / 8e12
            if e["l2_request_bytes"] is not None:
                fr["l2"] = e["l2_request_bytes"] / t / 34.5e12
            if e["instructions"] is not None:
                fr["issue"] = e["instructions"] / t / (simds * 2.4e9)
            e["fractions"] = fr
            if fr:
                e["bound"] = max(fr, key=fr.get)
                e["frac"] = fr[e["bound"]]
        e["tag"] = tag
        zj["kernels"][pre] = e
        fresh += 1
        if os.path.exists(st):
            shutil.copy(st, os.path.join(dst, f"{tag}_kernel_stats_{pre}.csv"))
        zmd += (f"| {kname} | {algo} | {what} | {e['launches']} | {(a or 0) / 1e3:.1f} | {e.get('hbm_bytes') or 0:.0f} | {e.get('l2_request_bytes') or 0:.0f} | "
                f"{e.get('instructions') or 0:.0f} | {(e.get('wait_fraction') or 0):.2f} | {(e.get('lds_bank_conflict_per_active_lds_cycle') or 0):.2f} | "
                + " / ".join(f"{k2} {v2:.4f}" for k2, v2 in (e.get('fractions') or {}).items()) + " |\n")
    if fresh:
        json.dump(zj, open(os.path.join(dst, "pmc_zoned.json"), "w"), indent=1)
        md += ("## the zone-aware and minimal-fragmentation kernels, one instantiation per profiled command (`tools/profile_cmd.py`)\n\n"
               "| kernel | packer | what | launches | avg us | HBM B | L2 request B | instructions | wait | LDS conflict / active | fractions |\n"
               "|---|---|---|---|---|---|---|---|---|---|---|\n" + zmd + "\n" + zj["note"] + "\n\n")

    # ---- the full run (every chain kernel) + its LDS counter set
    stats = os.path.join(src, "stats", "stats_kernel_stats.csv")
    if os.path.exists(stats):
        shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
        md += ("## kernel stats of the full bench run (`--steps 50 --windows 3 --filter-calls 30 --worker-sets 0`, extras on)\n\n" +
               _stats_table(stats, None) + "\n")
    lds = collections.OrderedDict()
    for r in _rows(src, "pmc_lds"):
        lds.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    if lds:
        md += "## LDS counter set of the full run (mean per launch)\n\n| kernel | counter | launches | mean |\n|---|---|---|---|\n"
        for (k, c), v in lds.items():
            pmc_csv.append(["full", k, c, len(v), f"{sum(v) / len(v):.3f}", f"{min(v):.3f}", f"{max(v):.3f}"])
            if "fit_" in k:
                md += f"| {k} | {c} | {len(v)} | {sum(v) / len(v):.1f} |\n"
        md += "\n"
    hf = os.path.join(src, "host_filter.txt")
    if os.path.exists(hf):
        shutil.copy(hf, os.path.join(dst, f"{tag}_host_filter.txt"))
        md += "## the whole Filter through the C++ mirror (`host_bench`)\n\n```\n" + open(hf, errors="replace").read() + "\n```\n"
    with open(os.path.join(dst, f"{tag}_pmc.csv"), "w") as out:
        csv.writer(out).writerows(pmc_csv)
    open(os.path.join(dst, f"{tag}_summary.md"), "w").write(md)
    print(md)


if __name__ == "__main__":
    main()
