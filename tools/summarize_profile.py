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
        if not os.path.exists(f):
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
                  "--no-cpu-baseline`, the same with `--headline-only`, and one `--pmc` pass per counter set (FETCH_SIZE, "
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
    print(open(os.path.join(dst, f"{tag}_summary.md")).read())


if __name__ == "__main__":
    main()
