"""Where a round of the resident worker goes: the instrumented build (tools/build_variants_fast.sh prof:"-DGF_WK_PROF", selected
with GANGFIT_LIB) sums shader cycles per phase over the rounds of ONE workgroup (set 0's first).  Run on the MI355X box."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import torch
import gangfit
from gangfit import workloads as wl

TIGHT = gangfit.GF_ALGO_TIGHTLY_PACK
dev = torch.device("cuda:0")
w = wl.headline(10000, 1000, seed=0x5EED0010)
s = w.snapshot
apps, total_k = gangfit.with_offsets(gangfit.make_apps(w.drv, w.exe, w.k, w.flags))
d_apps = torch.from_numpy(apps.view(np.uint8).copy()).to(dev)
outs = [(torch.zeros(len(apps) * 16, dtype=torch.uint8, device=dev), torch.zeros(total_k + 1, dtype=torch.int32, device=dev)) for _ in range(8)]
ctx = gangfit.Context(0)
ctx.set_snapshot(s.avail, s.sched)
ctx.set_orders(s.driver_order, s.exec_order)
names = ["probe/arrive", "barrier", "first record", "decisions", "stores issued", "drain"]
for K in (20, 2000):
    arr = ctx.worker_batches([(len(apps), d_apps.data_ptr(), outs[i % 8][0].data_ptr(), outs[i % 8][1].data_ptr(), total_k) for i in range(K)])
    for _ in range(2):
        ctx.worker_submit_prepared(TIGHT, arr); ctx.worker_stop(); torch.cuda.synchronize()
    ctx.scan_stats(enable=True, reset=True)
    ctx.worker_submit_prepared(TIGHT, arr); ctx.worker_stop(); torch.cuda.synchronize()
    lp_looks, lp_ticks = ctx.scan_stats(enable=False, reset=False)
    rounds = ctx.last_fifo_clock[0]
    lp = ctx.last_fifo_clock[1]
    print(f"   the leader: {lp_looks} looks that found tickets, {lp_ticks / max(1, lp_looks) / 100:.1f} us from a look to the end of its relay, "
          f"{(lp & 0xFFFFFFFF) / max(1, lp_looks):.1f} tickets per look (most: {lp >> 32})")
    w0 = ctx.last_fifo_phases
    out = np.zeros(12, dtype=np.uint64)
    ctx._check(ctx._lib.gf_chain_profile(ctx._h, gangfit._native.ptr(out)))
    w5 = [int(out[5 + i]) for i in range(5)] + [int(out[0])]
    sets, bps = ctx.worker_geometry()
    print(f"K = {K}: {sets} sets x {bps} workgroups, {rounds} rounds of the probed workgroup; shader cycles per round")
    for nm, a, b in zip(names, w0, w5):
        print(f"   {nm:14s} wavefront 0 {a / max(1, rounds):9.0f}   wavefront 5 {b / max(1, rounds):9.0f}")
    print(f"   over all workgroups, wavefront 0: waiting for a ticket {(1000 - int(out[2])) / 10:.1f} .. {int(out[1]) / 10:.1f} % of the rounds; decisions per round {(1 << 30) - int(out[4])} .. {int(out[3])} cycles")
    hw = int(out[10]) | (int(out[11]) << 32)
    print(f"   probed workgroup, wavefront 0: the next ticket was there when the round began in {hw & 0xFFFFFFFF} of {rounds} rounds; {hw >> 32} probes in the probe loop")
    print(f"   tickets: {int(out[0])} completed, {int(out[5]) / max(1, int(out[0])) / 100:.1f} us on average from the FIRST workgroup's share to the ticket's completion")
    print(f"   {'sum':14s} wavefront 0 {sum(w0) / max(1, rounds):9.0f}   wavefront 5 {sum(w5) / max(1, rounds):9.0f}   (2.4 GHz: {sum(w0) / max(1, rounds) / 2400:.2f} us)")
ctx.close()
