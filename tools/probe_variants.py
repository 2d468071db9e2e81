"""One line per build variant (GANGFIT_LIB): kernel time of the headline / config 3 / config 4 independent batches (graph of
20 launches, HIP events) with a checksum of the answers, and the cold chains of every packer (best of N host calls).
    for v in "" nopair w8; do GANGFIT_LIB=$PWD/k8s-spark-scheduler_amd/variants/libgangfit_$v.so python tools/probe_variants.py; done
"""
import os, sys, time, zlib
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import torch
import gangfit
from gangfit import workloads as wl

what = set((sys.argv[1] if len(sys.argv) > 1 else "ind,chain").split(","))
dev = torch.device("cuda:0")
out = [os.path.basename(os.environ.get("GANGFIT_LIB", "default"))]


def ind(name, w, algo, steps=20):
    ctx = gangfit.Context(0)
    ctx.set_snapshot(w.snapshot.avail, w.snapshot.sched)
    ctx.set_orders(w.snapshot.driver_order, w.snapshot.exec_order)
    a, tk = gangfit.with_offsets(gangfit.make_apps(w.drv, w.exe, w.k, w.flags))
    d_a = torch.from_numpy(a.view(np.uint8).copy()).to(dev)
    d_r = torch.zeros(len(a) * 16, dtype=torch.uint8, device=dev)
    d_e = torch.zeros(tk + 1, dtype=torch.int32, device=dev)
    f = lambda: ctx.fit_batch_dev(0, algo, len(a), d_a.data_ptr(), d_r.data_ptr(), d_e.data_ptr(), tk)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    ctx.graph_begin(0)
    for _ in range(steps):
        f()
    g = ctx.graph_end(0)
    ts = []
    for _ in range(7):
        torch.cuda.synchronize()
        ctx.timer_begin(0)
        ctx.graph_launch(g, 0)
        ts.append(ctx.timer_end() / steps * 1e3)
    ctx.graph_destroy(g)
    crc = zlib.crc32(d_r.cpu().numpy().tobytes()) ^ zlib.crc32(d_e.cpu().numpy().tobytes())
    out.append(f"{name} {sorted(ts)[3]:.2f}us crc {crc:08x}")
    ctx.close()


if "ind" in what:
    ind("headline:tight", wl.headline(10000, 1000), 0)
    w3 = wl.config(3)
    ind("c3:tight", w3, 0)
    ind("c3:even", w3, 1)
    ind("c3:minfrag", w3, 2, steps=5)
    ind("c4:tight", wl.config(4), 0, steps=10)
if "chain" in what:
    w = wl.headline(10000, 1000)
    s = w.snapshot
    zone3 = (wl.splitmix64(0xA3, len(s.avail), 9) % np.uint64(3)).astype(np.uint32)
    apps = gangfit.make_apps(w.drv, w.exe, w.k, w.flags)
    ctx = gangfit.Context(0, options={"chain_cache": 0})
    ctx.set_snapshot(s.avail, s.sched)
    ctx.set_zones(zone3)
    zo = wl.reference_node_order(s.avail, zone3)
    for algo, name, reps, order in ((0, "tight", 7, None), (1, "even", 5, None), (2, "minfrag", 3, None),
                                    (4, "azmajor:saz-tight", 4, zo), (3, "azmajor:az-aware", 3, zo), (5, "azmajor:saz-minfrag", 3, zo)):
        if order is None:
            ctx.set_orders(s.driver_order, s.exec_order)
        else:
            ctx.set_orders(order, order)
        r = ctx.fit_batch(1, algo, apps)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = ctx.fit_batch(1, algo, apps)
            ts.append((time.perf_counter() - t0) * 1e3)
        crc = zlib.crc32(r.results.tobytes()) ^ zlib.crc32(r.exec_nodes.tobytes())
        out.append(f"{name} {min(ts):.3f}ms crc {crc:08x}")
    ctx.close()
print("  ".join(out), flush=True)
