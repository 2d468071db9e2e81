"""Resident worker of the independent batch: window time for K tickets (launch + K tickets + stop), steady-state ticket rate,
and the blocking host call (gf_worker_fit) against gf_fit_batch.  Run on the MI355X box."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import torch
import gangfit
from gangfit import workloads as wl

IND, TIGHT = gangfit.GF_MODE_INDEPENDENT, gangfit.GF_ALGO_TIGHTLY_PACK
dev = torch.device("cuda:0")
w = wl.headline(10000, 1000, seed=0x5EED0010)
s = w.snapshot
apps, total_k = gangfit.with_offsets(gangfit.make_apps(w.drv, w.exe, w.k, w.flags))
d_apps = torch.from_numpy(apps.view(np.uint8).copy()).to(dev)
NOUT = 8
outs = [(torch.zeros(len(apps) * 16, dtype=torch.uint8, device=dev), torch.zeros(total_k + 1, dtype=torch.int32, device=dev)) for _ in range(NOUT)]


def med(x):
    x = sorted(x)
    return x[len(x) // 2]


for sets in (1, 2, 3):
    for bps in (64,):
        ctx = gangfit.Context(0, options={"worker_sets": sets, "worker_blocks_per_set": bps})
        ctx.set_snapshot(s.avail, s.sched)
        ctx.set_orders(s.driver_order, s.exec_order)
        for K in (20, 200, 2000):
            arr = ctx.worker_batches([(len(apps), d_apps.data_ptr(), outs[i % NOUT][0].data_ptr(), outs[i % NOUT][1].data_ptr(), total_k) for i in range(K)])
            walls, steady = [], []
            for rep in range(8):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                first = ctx.worker_submit_prepared(TIGHT, arr)
                ctx.worker_wait(first, K)
                t1 = time.perf_counter()
                ctx.worker_stop()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                if rep:
                    walls.append(t2 - t0)
                    steady.append(t1 - t0)
            print(f"sets {sets} bps {bps} K {K}: window {med(walls)*1e6:8.1f} us = {med(walls)/K*1e6:6.2f} us/batch = {len(apps)*K/med(walls)/1e6:7.1f} M/s;"
                  f" submit+wait only {med(steady)/K*1e6:6.2f} us/batch", flush=True)
        # resident across windows (no stop): what a host that keeps posting sees
        K = 200
        arr = ctx.worker_batches([(len(apps), d_apps.data_ptr(), outs[i % NOUT][0].data_ptr(), outs[i % NOUT][1].data_ptr(), total_k) for i in range(K)])
        ctx.set_option("worker_idle_us", 100000)
        ts = []
        for rep in range(10):
            t0 = time.perf_counter()
            first = ctx.worker_submit_prepared(TIGHT, arr)
            ctx.worker_wait(first, K)
            ts.append(time.perf_counter() - t0)
        print(f"sets {sets}: resident, {K} tickets per group: {med(ts)/K*1e6:6.2f} us/batch = {len(apps)*K/med(ts)/1e6:7.1f} M/s", flush=True)
        # lone ticket latency on a resident worker
        one = ctx.worker_batches([(len(apps), d_apps.data_ptr(), outs[0][0].data_ptr(), outs[0][1].data_ptr(), total_k)])
        ts = []
        for rep in range(200):
            t0 = time.perf_counter()
            first = ctx.worker_submit_prepared(TIGHT, one)
            ctx.worker_wait(first, 1)
            ts.append(time.perf_counter() - t0)
        print(f"sets {sets}: lone ticket on a resident worker p50 {med(ts)*1e6:6.2f} us", flush=True)
        ha = gangfit.make_apps(w.drv, w.exe, w.k, w.flags)
        for name, fn in (("gf_worker_fit", lambda: ctx.worker_fit(TIGHT, ha)), ("gf_fit_batch", lambda: ctx.fit_batch(IND, TIGHT, ha))):
            for _ in range(20):
                fn()
            ts = []
            for rep in range(300):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            print(f"sets {sets}: {name} (host arrays, blocking, through ctypes) p50 {med(ts)*1e6:6.2f} us", flush=True)
        ctx.close()
