"""Register / scratch / occupancy table of every kernel of libgangfit (hipcc -Rpass-analysis=kernel-resource-usage on the two
kernel files; compiles only, runs anywhere hipcc does).   python tools/kernel_resources.py [extra hipcc flags] > profiles/<tag>_kernel_resources.txt
V = VGPRs, S = SGPRs, scr = scratch bytes per lane, occ = wavefronts per SIMD, sspill / vspill = spilled SGPRs / VGPRs, lds = static LDS bytes."""
import os, re, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "k8s-spark-scheduler_amd", "csrc")
INC = os.path.join(REPO, "include")


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"^void gangfit::\(anonymous namespace\)::|^void gangfit::|\(.*$", "", o) for o in out[: len(names)]]


def table(src, extra):
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", INC, "-I", CSRC, "-Rpass-analysis=kernel-resource-usage",
               *extra, "-c", os.path.join(CSRC, src), "-o", os.path.join(tmp, "o.o")]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.split("\n"):
        m = re.search(r"remark: .*Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        for key, pat in (("S", r"TotalSGPRs: (\d+)"), ("V", r"\bVGPRs: (\d+)"), ("scr", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None and "remark" in line:
                cur[key] = int(m.group(1))
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        print(f"{n:72s} V {r.get('V', 0):3d} S {r.get('S', 0):3d} scr {r.get('scr', 0):4d} occ {r.get('occ', 0)} sspill {r.get('sspill', 0):3d} vspill {r.get('vspill', 0):3d} lds {r.get('lds', 0)}")


if __name__ == "__main__":
    extra = sys.argv[1:]
    print(f"# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage {' '.join(extra)}".rstrip())
    print("# V = VGPRs, S = SGPRs, scr = scratch bytes per lane, occ = wavefronts per SIMD, sspill / vspill = spilled SGPRs / VGPRs, lds = static LDS bytes")
    for src in ("gangfit_kernels.hip", "gangfit_snapshot.hip"):
        print(f"# ---- {src}")
        table(src, extra)
