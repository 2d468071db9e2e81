#!/usr/bin/env python
"""Turn the rocprofv3 CSVs of tools/profile_round.sh (gpurun_out/prof_<tag>/) into the tracked summaries under profiles/:
  profiles/<tag>_kernel_stats.csv   — rocprofv3 --kernel-trace --stats summary (verbatim)
  profiles/<tag>_pmc.csv            — per-kernel averages of every PMC pass (one counter set per pass)
  profiles/<tag>_summary.md         — both, human readable, with the gfx950 FETCH_SIZE correction applied
  profiles/pmc_traffic.json         — HBM bytes per launch of the dominant kernels (read by bench.py -> roofline.traffic)
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes for
wide coalesced reads (MI355X_MICROARCH.md, HBM section), so read bytes = 2 x FETCH_SIZE x 1024 is the upper estimate
and 1 x the lower one; both are recorded, `traffic` uses the corrected (2x) figure.
"""
import collections
import csv
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
    src = os.path.join(REPO, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(REPO, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = os.path.join(src, "stats", "stats_kernel_stats.csv")
    shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    krows = list(csv.DictReader(open(stats)))
    pmc = collections.OrderedDict()
    for d in sorted(os.listdir(src)):
        f = os.path.join(src, d, "pmc_counter_collection.csv")
        if not os.path.exists(f) or not d.startswith("pmc_"):  # chain_* / c3_* passes: separate commands, summarised below
            continue
        for r in csv.DictReader(open(f)):
            key = (r["Kernel_Name"], r["Counter_Name"])
            pmc.setdefault(key, []).append(float(r["Counter_Value"]))
        meta = {(r["Kernel_Name"]): (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"],
                                     r["SGPR_Count"]) for r in csv.DictReader(open(f))}
        pmc.setdefault("_meta", {}).update(meta)
    meta = pmc.pop("_meta", {})
    with open(os.path.join(dst, f"{tag}_pmc.csv"), "w") as out:
        w = csv.writer(out)
        w.writerow(["kernel", "counter", "launches", "mean", "min", "max"])
        for (k, c), v in pmc.items():
            w.writerow([k, c, len(v), f"{sum(v) / len(v):.3f}", f"{min(v):.3f}", f"{max(v):.3f}"])
    traffic = {}
    for (k, c), v in pmc.items():
        if c in ("FETCH_SIZE", "WRITE_SIZE"):
            traffic.setdefault(k, {})[c] = sum(v) / len(v)
    tj = {"tag": tag, "note": "bytes per launch; read = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB"}
    for k, t in traffic.items():
        if "fit_" not in k:
            continue
        rd, wr = t.get("FETCH_SIZE", 0.0) * 1024, t.get("WRITE_SIZE", 0.0) * 1024
        tj[k] = {"fetch_bytes_raw": rd, "fetch_bytes_corrected": 2 * rd, "write_bytes": wr, "hbm_bytes": 2 * rd + wr}
    if "fit_independent_kernel" in tj:
        tj["fit_independent_tight_headline_bytes_per_launch"] = tj["fit_independent_kernel"]["hbm_bytes"]
    json.dump(tj, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
    with open(os.path.join(dst, f"{tag}_summary.md"), "w") as out:
        out.write(f"# rocprofv3 summary — {tag}\n\nCommand: `tools/profile_round.sh {tag}` on one MI355X (gfx950), "
                  "i.e. `rocprofv3 --kernel-trace --stats -- python bench.py --steps 50 --warmup 5 --windows 3 --filter-calls 30 "
                  "--no-cpu-baseline --worker-sets 0` (the launch path: every batch a dispatch row), the same with `--headline-only`, and one `--pmc` pass per counter set (FETCH_SIZE, "
                  "WRITE_SIZE and the L2 pair on the headline alone; the SQ and LDS sets on the full run so that the FIFO chain "
                  "kernels are covered).\n\n")
        hstats = os.path.join(src, "stats_headline", "stats_kernel_stats.csv")
        if os.path.exists(hstats):
            shutil.copy(hstats, os.path.join(dst, f"{tag}_kernel_stats_headline.csv"))
            out.write("## the headline alone (`--headline-only`: every dispatch of fit_independent_kernel is a headline launch)\n\n")
            out.write("| kernel | calls | avg ns | min ns | max ns |\n|---|---|---|---|---|\n")
            for r in csv.DictReader(open(hstats)):
                if "fit_independent" in r["Name"] or "empty_kernel" in r["Name"]:
                    out.write(f"| {r['Name']} | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} |\n")
            out.write("\n")
        wstats = os.path.join(src, "stats_worker", "stats_kernel_stats.csv")
        if os.path.exists(wstats):
            shutil.copy(wstats, os.path.join(dst, f"{tag}_kernel_stats_worker.csv"))
            wj = _bench_json(os.path.join(src, "stats_worker.log"))
            out.write("## the headline through the resident worker (`bench.py --steps 200 --headline-only`, worker on)\n\n"
                      "One dispatch of `fit_worker_kernel` per window (it serves the window's 200 tickets and leaves); the "
                      "launch-path windows of the same run appear as `fit_independent_kernel` rows.\n\n")
            out.write("| kernel | calls | avg ns | min ns | max ns |\n|---|---|---|---|---|\n")
            for r in csv.DictReader(open(wstats)):
                if "fit_worker" in r["Name"] or "fit_independent" in r["Name"]:
                    out.write(f"| {r['Name']} | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} |\n")
            if wj:
                out.write(f"\nbench line of that run: value {wj.get('value', 0) / 1e6:.1f} M decisions/s, "
                          f"{wj.get('ms_per_step', 0) * 1e3:.2f} us per step; launch path "
                          f"{wj.get('timing', {}).get('launch_path', {}).get('value', 0) / 1e6:.1f} M/s\n")
            out.write("\n")
        out.write("## kernel stats (all kernels of the full bench run)\n\n")
        out.write("| kernel | calls | avg ns | min ns | max ns | % |\n|---|---|---|---|---|---|\n")
        for r in krows:
            out.write(f"| {r['Name']} | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | "
                      f"{float(r['Percentage']):.3f} |\n")
        out.write("\nNote: in the full run `fit_independent_kernel` covers every launch (tightly-pack, distribute-evenly, "
                  "nominal and congested batches, config 3 and 4, the host-entry batches that read their records over PCIe); the "
                  "headline-only table above isolates the headline launches.  FETCH_SIZE / WRITE_SIZE / TCC passes ran "
                  "`--headline-only`; the SQ / LDS passes ran the full bench (means over all launches of a kernel).\n\n"
                  "## PMC passes (mean per launch)\n\n")
        out.write("| kernel | counter | launches | mean | min | max |\n|---|---|---|---|---|---|\n")
        for (k, c), v in pmc.items():
            if "fit_" not in k and "translate" not in k:
                continue
            out.write(f"| {k} | {c} | {len(v)} | {sum(v) / len(v):.1f} | {min(v):.1f} | {max(v):.1f} |\n")
        out.write("\nKernel resources (grid, workgroup, LDS, VGPR, SGPR): " +
                  "; ".join(f"{k}: {v}" for k, v in meta.items() if "fit_" in k) + "\n\n")
        out.write("## HBM traffic per launch\n\n```json\n" + json.dumps(tj, indent=1) + "\n```\n")
    extra = chain_and_config3(tag, src, dst)
    with open(os.path.join(dst, f"{tag}_summary.md"), "a") as out:
        out.write(extra)
    print(open(os.path.join(dst, f"{tag}_summary.md")).read())


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


def _pass(src, d):
    f = os.path.join(src, d, "pmc_counter_collection.csv")
    return list(csv.DictReader(open(f))) if os.path.exists(f) else []


def chain_and_config3(tag, src, dst):
    """profiles/pmc_chain.json (instructions per application of the plain FIFO chain: `--fifo-protocols cold`, every launch a
    full replay of the headline chain) and profiles/pmc_config3.json (HBM bytes / L2 requests per config-3 launch, per packer),
    plus their sections of the summary."""
    md = ""
    # ---- the chain
    rows = [r for r in _pass(src, "chain_sq") if r["Kernel_Name"].startswith("fit_fifo_solo_kernel")]
    if rows:
        per = collections.OrderedDict()
        for r in rows:
            per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        mean = {c: sum(v) / len(v) for c, v in per.items()}
        bj = _bench_json(os.path.join(src, "chain_sq.log")) or {}
        n_apps = ((bj.get("roofline") or {}).get("fifo_chain") or {}).get("applications_per_chain") or 1000
        insts = sum(mean.get(c, 0.0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"))
        cj = {"tag": tag, "kernel": "fit_fifo_solo_kernel", "launches": len(next(iter(per.values()))),
              "applications_per_chain": n_apps, "per_launch": mean,
              "fit_fifo_solo_instructions_per_app": insts / n_apps,
              "note": "VALU + SALU + LDS + SMEM instructions of ALL sixteen wavefronts per application (the fifteen helpers only run "
                      "the prologue, the checkpoint dumps and the epilogue); every launch is a full replay of the headline chain "
                      "(bench.py --fifo-protocols cold)"}
        kst = os.path.join(src, "stats_chain", "stats_kernel_stats.csv")
        if os.path.exists(kst):
            shutil.copy(kst, os.path.join(dst, f"{tag}_kernel_stats_chain.csv"))
            for r in csv.DictReader(open(kst)):
                if r["Name"].startswith("fit_fifo_solo_kernel"):
                    cj["kernel_avg_ns"] = float(r["AverageNs"])
                    cj["kernel_calls"] = int(r["Calls"])
        json.dump(cj, open(os.path.join(dst, "pmc_chain.json"), "w"), indent=1)
        md += ("\n## the plain FIFO chain alone (`--no-extras --fifo-protocols cold`: every launch replays the headline chain)\n\n"
               "```json\n" + json.dumps(cj, indent=1) + "\n```\n")
    # ---- config 3: the two packers are two instantiations of fit_independent_kernel (Kernel_Id in launch order)
    c3 = {"tag": tag, "note": "bytes per launch; read = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB; L2 requests "
                              "are 128-byte lines"}
    names = ["tightly_pack", "distribute_evenly"]
    for d, counters in (("c3_fetch", ["FETCH_SIZE"]), ("c3_write", ["WRITE_SIZE"]), ("c3_l2", ["TCC_HIT_sum", "TCC_MISS_sum"]),
                        ("c3_sq", ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAIT_ANY"])):
        rows = [r for r in _pass(src, d) if r["Kernel_Name"].startswith("fit_independent_kernel")]
        ids = []
        for r in rows:
            if r["Kernel_Id"] not in ids:
                ids.append(r["Kernel_Id"])
        for i, kid in enumerate(ids[:2]):
            for c in counters:
                v = [float(r["Counter_Value"]) for r in rows if r["Kernel_Id"] == kid and r["Counter_Name"] == c]
                if v:
                    c3.setdefault(names[i], {})[c] = sum(v) / len(v)
                    c3[names[i]]["launches"] = len(v)
    for nme in names:
        t = c3.get(nme)
        if not t:
            continue
        rd, wr = t.get("FETCH_SIZE", 0.0) * 1024, t.get("WRITE_SIZE", 0.0) * 1024
        t["hbm_bytes"] = 2 * rd + wr
        t["l2_request_bytes"] = 128 * (t.get("TCC_HIT_sum", 0.0) + t.get("TCC_MISS_sum", 0.0))
    kst = os.path.join(src, "stats_config3", "stats_kernel_stats.csv")
    if os.path.exists(kst):
        shutil.copy(kst, os.path.join(dst, f"{tag}_kernel_stats_config3.csv"))
    if any(n in c3 for n in names):
        json.dump(c3, open(os.path.join(dst, "pmc_config3.json"), "w"), indent=1)
        md += ("\n## BASELINE config 3 alone (`bench.py --config3-only`: 10 000 nodes x 10 000 apps, both packers)\n\n```json\n" +
               json.dumps(c3, indent=1) + "\n```\n")
    return md


if __name__ == "__main__":
    main()
