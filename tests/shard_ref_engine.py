"""TEST INFRASTRUCTURE — a numpy stand-in for HipShardEngine (k8s-spark-scheduler_amd/gangfit/sharded.py) so that the
N > 1 control flow (exchanges, prefix bookkeeping, finish) runs on CPU under gloo.  It restates the four per-shard steps
of csrc/gangfit_shard.inc from their specification (SURVEY.md section 8 "array restatement" + 8e) with exact, never
early-stopped sums; its output is checked against the oracle's unsharded answer, never used by the product."""
import numpy as np
import torch

NO_NODE = 0xFFFFFFFF
TIGHT, EVEN = 0, 1
APP_DTYPE = np.dtype([("drv", "<i8", (3,)), ("exe", "<i8", (3,)), ("k", "<i4"), ("flags", "<u4"), ("exec_off", "<u8")],
                     align=True)


def _cap(avail, base, exe, k):
    """min over dims of: 0 if avail-base < 0; inf if exe == 0; floor((avail-base)/exe) — clamped to k."""
    a = avail - base
    out = np.full(len(a), k, dtype=np.int64)
    for j in range(3):
        cj = np.where(a[:, j] < 0, 0, k if exe[j] == 0 else np.minimum(a[:, j] // max(int(exe[j]), 1), k))
        out = np.minimum(out, cj)
    return out


class RefShardEngine:
    def __init__(self, avail, merged_order, xflag, dflag, shard, n_shards):
        self.avail = np.asarray(avail, dtype=np.int64).reshape(-1, 3)
        self.order = np.asarray(merged_order, dtype=np.int64)
        self.x = np.asarray(xflag, dtype=bool)
        self.d = np.asarray(dflag, dtype=bool)
        self.shard, self.n_shards = shard, n_shards
        xc = (len(self.order) + 63) // 64
        self.lo = min(len(self.order), 64 * (xc * shard // n_shards))
        self.hi = min(len(self.order), 64 * (xc * (shard + 1) // n_shards))
        self.tab = self.avail[self.order]  # slot-ordered

    def stream_context(self):
        import contextlib

        return contextlib.nullcontext()

    def upload_apps(self, apps_off):
        return np.ascontiguousarray(apps_off).view(APP_DTYPE)

    def _caps0(self, app):
        c = _cap(self.tab, np.zeros(3, dtype=np.int64), app["exe"], int(app["k"]))
        return np.where(self.x, c, 0)

    def partials(self, algo, apps, n_apps):
        out = np.zeros((n_apps, 2), dtype=np.int64)
        for a in range(n_apps):
            if apps[a]["k"] == 0:
                continue
            c = self._caps0(apps[a])[self.lo:self.hi]
            out[a] = (c.sum(), (c >= 1).sum())
        return torch.from_numpy(out)

    def drivers(self, algo, apps, n_apps, all_part):
        all_part = all_part.numpy()
        out = np.zeros((n_apps, 4), dtype=np.int32)
        out[:, 0] = -1  # GF_NO_NODE as int32
        for a in range(n_apps):
            app = apps[a]
            k = int(app["k"])
            S = int(all_part[:, a, 0].sum())
            c0 = self._caps0(app)
            c1 = np.where(self.x, _cap(self.tab, app["drv"], app["exe"], k), 0)
            fits = self.d & (self.tab >= app["drv"]).all(axis=1)
            total = S - c0 + c1
            ok = fits & (total >= k)
            idx = np.nonzero(ok[self.lo:self.hi])[0]
            if len(idx):
                p = self.lo + int(idx[0])
                out[a] = (p, c1[p] - c0[p], int(c1[p] >= 1) - int(c0[p] >= 1), 0)
        return torch.from_numpy(out)

    def _sums(self, all_part, all_drv, a, shard):
        pos = all_drv[:, a, 0].astype(np.int64) & 0xFFFFFFFF
        owner = int(np.argmin(pos))
        if pos[owner] == NO_NODE:
            return None
        cap = all_part[:, a, 0].copy()
        fit = all_part[:, a, 1].copy()
        cap[owner] += all_drv[owner, a, 1]
        fit[owner] += all_drv[owner, a, 2]
        return int(pos[owner]), int(cap[:shard].sum()), int(fit[:shard].sum()), int(fit.sum())

    def emit(self, algo, apps, n_apps, all_part, all_drv, half):
        all_part, all_drv = all_part.numpy(), all_drv.numpy()
        res = np.zeros(n_apps, dtype=[("has_capacity", "<i4"), ("driver_node", "<u4"), ("exec_len", "<u4"),
                                      ("evaluated", "<u4")])
        exec2 = np.zeros(2 * half, dtype=np.int32)
        for a in range(n_apps):
            app = apps[a]
            k, off = int(app["k"]), int(app["exec_off"])
            g = self._sums(all_part, all_drv, a, self.shard)
            if g is None:
                res[a] = (0, NO_NODE, 0, 1)
                continue
            pos, before_cap, before_fit, total_fit = g
            res[a] = (1, self.order[pos], k, 1)
            if k == 0:
                continue
            base = np.zeros((len(self.order), 3), dtype=np.int64)
            base[pos] = app["drv"]
            caps = np.where(self.x, _cap(self.tab, base, app["exe"], k), 0)
            if algo == TIGHT:
                taken = before_cap
                for j in range(self.lo, self.hi):
                    t = min(int(caps[j]), max(0, k - taken))
                    exec2[off + min(taken, k): off + min(taken, k) + t] = self.order[j] + 1
                    taken += int(caps[j])
            else:
                p = before_fit
                multipass = total_fit < k
                for j in range(self.lo, self.hi):
                    if caps[j] >= 1:
                        if p < k:
                            exec2[off + p] = self.order[j] + 1
                            if multipass:
                                exec2[half + off + p] = caps[j]
                        p += 1
        return torch.from_numpy(res.view(np.uint8).copy()), torch.from_numpy(exec2)

    def finish(self, algo, apps, n_apps, all_part, all_drv, res, exec2, half):
        all_part, all_drv = all_part.numpy(), all_drv.numpy()
        r = res.numpy().view([("has_capacity", "<i4"), ("driver_node", "<u4"), ("exec_len", "<u4"), ("evaluated", "<u4")])
        ex = exec2.numpy()
        for a in range(n_apps):
            if not r[a]["has_capacity"]:
                continue
            k, off = int(apps[a]["k"]), int(apps[a]["exec_off"])
            m1 = k
            if algo == EVEN:
                total_fit = self._sums(all_part, all_drv, a, 0)[3]
                m1 = min(k, total_fit)
            ex[off:off + m1] -= 1
            p, rr = m1, 2
            while p < k:
                for i in range(m1):
                    if ex[half + off + i] >= rr and p < k:
                        ex[off + p] = ex[off + i]
                        p += 1
                rr += 1
