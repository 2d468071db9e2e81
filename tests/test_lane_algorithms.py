"""CPU models of two lane-parallel device routines of the zone-aware chains (csrc/gangfit_fifo_zoned.inc), checked against the
sequential definitions they replace.  The GPU parity tests compare the kernels' float64 results with the oracle bit for bit;
these models document WHY the routines are exact, step by step, and run without a GPU.

  wave_serial_sum_runs   the slice-order sum of ComputeAvgPackingEfficiency (efficiency.go:114-156) as a systolic pass
  zoned_choose_best      chooseBestResult (single_az.go:75-97) as a row max-scan
  narrow_magic           floor(a / e) as a multiplication (gangfit_fifo_common.inc), the chain kernels' capacity arithmetic
  zoned_choose_bounded   chooseBestResult from (tree sum, error bound) pairs: never a different winner than the exact sums
  chunk_gcd_rounds       the gcd of a 64-slot chunk by candidate rounds (csrc/gangfit_snapshot.hip: finalize_slots_kernel) and the
                         two-operand gcd with the common power of two set aside (gcd_fast), against math.gcd
  checkpoint_lookup      the source checkpoint of a delta-format overlay, lane = checkpoint (csrc/gangfit_fifo_common.inc:
                         chain_prologue_kernel), against the walk back through the checkpoints
  minfrag_histogram      minimalFragmentation (minimal_fragmentation.go:59-137) decided on the histogram of the capacities + one
                         emission pass (csrc/gangfit_minfrag.inc: wave_minfrag_hist), against the walk over the sorted list
  minfrag_team           the same decision by a TEAM of wavefronts (team_minfrag_hist): quarters of the order's chunks, private count /
                         first-slot rows, the prefix of the quarters before as the starting ranks of the emission pass
"""
import numpy as np
import pytest

WAVE = 64


def sequential_sum(carry, values, counts):
    acc = np.float64(carry)
    for v, c in zip(values, counts):
        for _ in range(int(c)):
            acc = np.float64(acc + np.float64(v))
    return acc


def systolic_sum(carry, values, counts):
    """The device routine, lane by lane: lane t owns the steps [start_t, start_t + c_t); at its first step it takes the
    running sum of lane t - 1 (a shift by one lane), and it adds its value at EVERY step — what it holds outside its
    window is never read.  Lanes beyond the runs add 0.0."""
    n = len(values)
    assert 1 <= n <= WAVE and all(c >= 1 for c in counts)
    v = np.zeros(WAVE, dtype=np.float64)
    c = np.zeros(WAVE, dtype=np.int64)
    v[:n] = values
    c[:n] = counts
    incl = np.cumsum(c)
    start = incl - c
    steps = int(incl[-1])
    acc = np.full(WAVE, np.float64(carry), dtype=np.float64)
    for q in range(steps):
        prev = np.concatenate(([np.float64(0.0)], acc[:-1]))      # DPP wave_shr:1, lane 0 reads 0
        take = (start == q)
        take[0] = False                                            # lane 0 starts from the carry it already holds
        acc = np.where(take, prev, acc) + v                        # every lane, every step
    return acc[n - 1]


@pytest.mark.parametrize("seed", range(40))
def test_systolic_sum_is_the_sequential_sum(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, WAVE + 1))
    # efficiencies are quotients of integers: values whose sums round differently in different orders
    values = rng.integers(0, 1 << 40, size=n) / rng.integers(1, 1 << 38, size=n)
    if seed % 5 == 0:
        values[rng.integers(0, n)] = 0.0
    counts = np.where(rng.random(n) < 0.8, 1, rng.integers(2, 40, size=n))
    carry = np.float64(0.0) if seed % 3 else np.float64(rng.random() * 7)
    want = sequential_sum(carry, values, counts)
    got = systolic_sum(carry, values, counts)
    assert got.tobytes() == want.tobytes()


def test_the_order_of_the_sum_matters():
    """float64 addition does not reassociate: the bit-for-bit checks above are not vacuous."""
    rng = np.random.default_rng(3)
    values = rng.integers(0, 1 << 40, size=48) / rng.integers(1, 1 << 38, size=48)
    counts = np.ones(48, dtype=np.int64)
    differing = sum(sequential_sum(0.0, values, counts).tobytes() != sequential_sum(0.0, rng.permutation(values), counts).tobytes()
                    for _ in range(20))
    assert differing > 0


def test_systolic_sum_chained_batches():
    """More than 64 runs: the result of one 64-lane batch is the carry of the next."""
    rng = np.random.default_rng(7)
    values = rng.integers(1, 1 << 30, size=150) / rng.integers(1, 1 << 28, size=150)
    counts = rng.integers(1, 4, size=150)
    want = sequential_sum(0.0, values, counts)
    acc = np.float64(0.0)
    for b in range(0, 150, WAVE):
        acc = systolic_sum(acc, values[b:b + WAVE], counts[b:b + WAVE])
    assert acc.tobytes() == want.tobytes()


def reference_choose_best(feas, mx):
    best, best_max = -1, np.float64(0.0)       # WorstAvgPackingEfficiency: every component 0 (efficiency.go:42-49)
    for c in range(len(feas)):
        if feas[c] and best_max < mx[c]:        # LessThan on Max (efficiency.go:36-40), strict: the first of equals stays
            best, best_max = c, mx[c]
    return best


def scan_choose_best(feas, mx):
    """The device routine: infeasible candidates hold a value that cannot win, an inclusive max-scan along the row leaves
    the largest in the last lane, the first feasible lane that holds it wins — if it is above 0."""
    n = len(feas)
    x = np.where(np.asarray(feas, dtype=bool), np.asarray(mx, dtype=np.float64), -1.0)
    row = np.full(16, -1.0)
    row[:n] = x
    for shift in (1, 2, 4, 8):                 # row_shr:n, lanes without a source keep their own value
        shifted = np.concatenate((row[:shift], row[:-shift]))
        row = np.maximum(row, shifted)
    top = row[15]
    if not top > 0.0:
        return -1
    for c in range(n):
        if feas[c] and mx[c] == top:
            return c
    return -1


@pytest.mark.parametrize("seed", range(200))
def test_choose_best_scan_matches_the_reference_loop(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 17))
    feas = rng.random(n) < 0.7
    pool = np.array([0.0, 0.25, 0.5, 0.5, 0.75, 0.9, 1.0, 1.5])   # ties and zeros on purpose
    mx = rng.choice(pool, size=n) if seed % 2 else rng.random(n)
    assert scan_choose_best(feas, mx) == reference_choose_best(feas, mx)


# ---------------------------------------------------------------------------------------------------------------------
# narrow_magic: floor(a / e) == mulhi(2 m, 2 a) >> l with l = ceil(log2 e), m = ceil(2^(30 + l) / e), for 0 <= a < 2^30, 0 < e < 2^30


def narrow_magic(e: int):
    l = 0 if e <= 1 else (e - 1).bit_length()
    m = ((1 << (30 + l)) + e - 1) // e
    assert m < (1 << 31)
    return 2 * m, l


def magic_div(a: int, mag: int, l: int) -> int:
    assert 0 <= a < (1 << 30) and 0 <= mag < (1 << 32)
    return ((mag * ((2 * a) & 0xFFFFFFFF)) >> 32) >> l


def test_narrow_magic_is_exact_on_edges_and_random_operands():
    rng = np.random.default_rng(2024)
    lim = (1 << 30) - 1
    divisors = [1, 2, 3, 5, 7, 1000, 1023, 1024, 1025, 4096, 65535, 65536, 65537, (1 << 29) - 1, 1 << 29, (1 << 29) + 1, lim]
    divisors += [int(x) for x in rng.integers(1, lim + 1, size=300)]
    divisors += [int(1 << int(b)) for b in rng.integers(0, 30, size=20)]
    for e in divisors:
        mag, l = narrow_magic(e)
        cands = [0, 1, e - 1, e, e + 1, 2 * e - 1, 2 * e, lim, lim - 1, (lim // e) * e, (lim // e) * e - 1]
        cands += [int(x) for x in rng.integers(0, lim + 1, size=40)]
        cands += [int(q) * e + d for q in rng.integers(0, max(lim // e, 1), size=20) for d in (-1, 0, 1)]
        for a in cands:
            if 0 <= a <= lim:
                assert magic_div(a, mag, l) == a // e, (a, e)


def test_narrow_magic_exhaustive_small():
    for e in range(1, 200):
        mag, l = narrow_magic(e)
        a = np.arange(0, 5000, dtype=np.int64)
        got = ((mag * ((2 * a) & 0xFFFFFFFF)) >> 32) >> l
        assert np.array_equal(got, a // e), e


# ---------------------------------------------------------------------------------------------------------------------
# zoned_choose_bounded: the winner from (value, bound) pairs is the exact chooser's winner whenever it answers at all


def choose_exact(feas, mx):
    """chooseBestResult: best_max starts at 0, a candidate wins on best_max < max: the FIRST of the largest, if above 0."""
    best, best_max = -1, 0.0
    for i, (f, m) in enumerate(zip(feas, mx)):
        if f and best_max < m:
            best, best_max = i, m
    return best


UNDECIDED = -2


def choose_bounded(feas, approx, err):
    cand = [i for i, f in enumerate(feas) if f]
    if not cand:
        return -1
    top = max(approx[i] for i in cand)
    w = next(i for i in cand if approx[i] == top)
    lo = approx[w] - err[w]
    if not lo > 0.0:
        if top == 0.0 and all(err[i] == 0.0 for i in cand):
            return -1
        return UNDECIDED
    for i in cand:
        if i != w and not (approx[i] + err[i] < lo) and not (err[w] == 0.0 and err[i] == 0.0):
            return UNDECIDED
    return w


def tree_sum(values):
    v = list(values) + [0.0] * (WAVE - len(values))
    v = np.array(v, dtype=np.float64)
    step = 1
    while step < WAVE:  # any pairing order: the bound only counts roundings
        v = v + np.concatenate((np.zeros(step), v[:-step]))
        step *= 2
    return np.float64(v[-1])


def test_bounded_chooser_never_disagrees_with_the_slice_order_sums():
    rng = np.random.default_rng(7)
    decided = undecided = 0
    for trial in range(3000):
        nz = int(rng.integers(1, 6))
        feas, exact, approx, err = [], [], [], []
        shared = None
        for z in range(nz):
            n = int(rng.integers(1, 40))
            counts = rng.integers(1, 6, size=n)
            values = rng.random(n) * rng.choice([1.0, 1e-3, 5.0])
            if trial % 7 == 0 and shared is not None:  # zones of equal nodes: exact ties
                counts, values = shared
            shared = (counts, values)
            k1 = int(counts.sum())
            ex = sequential_sum(0.0, values, counts) / np.float64(k1)
            ap = tree_sum(values * counts.astype(np.float64)) / np.float64(k1)
            bound = np.float64(4 * (k1 - 1) + 64) * 2.0 ** -53 * ap
            assert abs(ex - ap) <= bound / 2  # the proven bound, with the factor two the chooser's own roundings get
            feas.append(rng.random() < 0.9)
            exact.append(ex)
            approx.append(ap)
            err.append(bound)
        got = choose_bounded(feas, approx, err)
        want = choose_exact(feas, exact)
        if got == UNDECIDED:
            undecided += 1
        else:
            decided += 1
            assert got == want, (feas, exact, approx, err)
    assert decided > 1500 and undecided > 50  # both branches exercised (the ties are undecided by construction)


# ---------------------------------------------------------------------------------------------- reserved counts without a table
# fit_zoned_fused_kernel (csrc/gangfit_zones.inc, wave_avg_efficiency_runs) needs, for the average packing efficiency of a
# tightly-pack result, the number of executors on every node of the list (reserved[n] = count x exe, efficiency.go:79-103).
# Instead of counting them in a per-wavefront table it uses what tightlyPackExecutors guarantees (pack_tightly.go:45-61): every
# node of the list is one contiguous run filled to its capacity — computed with the driver reserved on the driver's node —,
# except the LAST node, which holds K minus the index of its first entry.  Checked here against the literal oracle.

def _cap(avail_row, exe, k):
    c = k
    for a, e in zip(avail_row, exe):
        if a < 0:
            return 0
        if e > 0:
            c = min(c, int(a) // int(e))
    return c


@pytest.mark.parametrize("seed", range(6))
def test_tightly_pack_counts_follow_from_capacities(seed):
    import os
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (repo, os.path.join(repo, "k8s-spark-scheduler_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from gangfit import workloads as wl
    from oracle import binding as ob

    w = wl.headline(200 + 37 * seed, 120, seed=0xC0FFEE + seed)
    s = w.snapshot
    apps = ob.make_apps(w.drv, w.exe, w.k, w.flags)
    ref = ob.fit_independent(0, s.avail, apps, s.driver_order, s.exec_order, closed_form=False)
    checked = multi = driver_listed = 0
    for a in np.nonzero(ref.results["has_capacity"])[0]:
        _, driver, nodes = ref.placement(int(a))
        k = int(w.k[a])
        if k == 0:
            continue
        nodes = [int(n) for n in nodes]
        # one contiguous run per node
        firsts = [i for i in range(k) if i == 0 or nodes[i - 1] != nodes[i]]
        assert len(set(nodes[i] for i in firsts)) == len(firsts)
        last, last_first = nodes[-1], firsts[-1]
        for i in firsts:
            n = nodes[i]
            true_count = nodes.count(n)
            row = s.avail[n].astype(np.int64).copy()
            if n == driver:
                row -= np.asarray(w.drv[a], dtype=np.int64)
            rule = (k - last_first) if n == last else _cap(row, w.exe[a], k)
            assert rule == true_count, (int(a), n, rule, true_count)
            checked += 1
            multi += true_count > 1
        driver_listed += driver in nodes
    assert checked > 100 and multi > 10 and driver_listed > 0


# ---------------------------------------------------------------------------------------------- unclamped quotients (cap_dim_full)
# minimal-fragmentation needs floor(a / e) unclamped (csrc/gangfit_minfrag.inc).  The device takes an f64 estimate a * (1 / e) and,
# when it is below 2^40, settles it with one exact multiply-subtract; anything larger goes to the plain 64-bit division.  Modelled
# here with IEEE doubles and a reciprocal that is off by up to 2^-44 relative (fast_rcp: the hardware estimate plus one Newton step).

def _cap_dim_full_model(a, e, rcp):
    if a < 0:
        return 0
    qf = float(a) * rcp
    if qf < 2.0 ** 40:
        q = int(qf)
        rem = a - q * e
        if rem < 0:
            q -= 1
        elif rem >= e:
            q += 1
        return q
    return a // e


@pytest.mark.parametrize("seed", range(8))
def test_unclamped_quotient_by_reciprocal_is_exact(seed):
    rng = np.random.default_rng(900 + seed)
    checked = big = 0
    for _ in range(20000):
        ebits = int(rng.integers(1, 62))
        e = max(1, int(rng.integers(0, 1 << ebits)))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            a = int(rng.integers(0, 1 << 62))
        elif kind == 1:  # just around a multiple of e
            a = min((1 << 62) - 1, int(rng.integers(0, 1 << 41)) * e + int(rng.integers(-1, 2)))
        elif kind == 2:
            a = int(rng.integers(0, e))
        else:  # right at the 2^40 switch
            a = min((1 << 62) - 1, (1 << 40) * e + int(rng.integers(-2, 3)))
        a = max(a, 0)
        for rel in (0.0, 2.0 ** -44, -(2.0 ** -44)):
            rcp = (1.0 / float(e)) * (1.0 + rel)
            assert _cap_dim_full_model(a, e, rcp) == a // e, (a, e, rel)
        checked += 1
        big += (a // e) >= (1 << 40)
    assert checked == 20000 and big > 100


# ------------------------------------------------------------------------------------------------ minimal-fragmentation, histogram form
# The sequential definition: minimal_fragmentation.go:59-137 on a list of (position, capacity) pairs, written as the reference
# writes it (stable sort by capacity, sort.Search, slices).  The device routine (wave_minfrag_hist) never sorts: ONE pass counts the
# capacities and keeps the first position of each, the whole walk is planned on the counts, and ONE more pass emits the runs of
# the drained levels (a node knows its rank among its level's nodes in priority order from a running count per level).

def _search(n, pred):
    lo, hi = 0, n
    while lo < hi:
        mid = (lo + hi) // 2
        if pred(mid):
            hi = mid
        else:
            lo = mid + 1
    return lo


def _internal_minfrag_reference(count, caps):
    """internalMinimalFragmentation (:93-137); caps = [(position, capacity)] sorted by capacity, stable."""
    caps = list(caps)
    out = []
    while caps:
        pos = _search(len(caps), lambda i: caps[i][1] >= count)
        if pos != len(caps):
            return out + [caps[pos][0]] * count, True
        mx = caps[-1][1]
        first = _search(len(caps), lambda i: caps[i][1] >= mx)
        cur = first
        while count >= mx and cur < len(caps):
            out += [caps[cur][0]] * mx
            count -= mx
            cur += 1
        if count == 0:
            return out, True
        caps = caps[:first] + caps[cur:]
    return None, False


def minfrag_reference(count, capacities):
    """minimalFragmentation (:59-91) for capacities[position] (0 = the node is filtered out)."""
    if count == 0:
        return [], True
    caps = sorted(((p, c) for p, c in enumerate(capacities) if c > 0), key=lambda pc: pc[1])  # sort.SliceStable
    if not caps:
        return None, False
    mx = caps[-1][1]
    if count < mx:
        target = (count + mx) // 2
        first = _search(len(caps), lambda i: caps[i][1] >= target)
        nodes, ok = _internal_minfrag_reference(count, caps[:first])
        if ok:
            return nodes, ok
    return _internal_minfrag_reference(count, caps)


def minfrag_histogram(count, capacities, bins=256):
    """The device routine, step by step (None = not applicable: a capacity beyond the last bin -> the walk serves the request)."""
    K = count
    if K == 0:
        return [], True
    # pass 1: counts and first positions (LDS atomics: order-free)
    cnt = [0] * bins
    first = [None] * bins
    for p, c in enumerate(capacities):
        if c >= bins:
            return None
        if c > 0:
            cnt[c] += 1
            if first[c] is None:
                first[c] = p
    S = sum(cnt[c] * min(c, K) for c in range(bins))
    if S < K:
        return None, False
    max_cap = max(c for c in range(bins) if cnt[c])
    top = bins
    if K < max_cap:
        target = (K + max_cap) // 2
        if sum(cnt[c] * min(c, K) for c in range(target)) >= K:
            top = target

    def smallest_at_least(need, below):
        return next((c for c in range(max(need, 1), below) if cnt[c]), None)

    def largest_below(below):
        return next((c for c in range(below - 1, 0, -1) if cnt[c]), 0)

    R = K
    cf = smallest_at_least(R, top)
    if cf is not None:
        return [first[cf]] * K, True
    take, base = {}, {}        # per drained level: nodes taken completely, where its runs start in the output
    last_pos, next_level, next_rank = None, None, 0
    while True:
        m = largest_below(top)
        assert 0 < m < R
        q = R // m
        drained = min(cnt[m], q)
        take[m], base[m] = drained, K - R
        R -= drained * m
        if R == 0:
            break
        if drained < cnt[m]:
            cf = smallest_at_least(R, m)
            if cf is not None:
                last_pos = first[cf]
            else:
                next_level, next_rank = m, drained
            break
        top = m
        cf = smallest_at_least(R, top)
        if cf is not None:
            last_pos = first[cf]
            break
    # pass 2: the runs, every node by its rank among its level's nodes in priority order
    out = [None] * K
    seen = {}
    for p, c in enumerate(capacities):
        if c in take or c == next_level:
            rank = seen.get(c, 0)
            seen[c] = rank + 1
            if rank < take.get(c, 0):
                for i in range(c):
                    out[base[c] + rank * c + i] = p
            if c == next_level and rank == next_rank:
                last_pos = p
    if R > 0:
        for i in range(R):
            out[K - R + i] = last_pos
    assert all(v is not None for v in out)
    return out, True


def minfrag_team(count, capacities, team=4, chunk=WAVE, bins=256):
    """team_minfrag_hist, wavefront by wavefront: wavefront r walks chunks [r * per, (r + 1) * per) of the order in BOTH passes and
    writes only its own rows; what crosses between the wavefronts is the sum of the count rows, the minimum of the first-slot rows
    and — per wavefront — the sum of the rows of the quarters before it.  Everything else is minfrag_histogram's plan."""
    K = count
    if K == 0:
        return [], True
    n = len(capacities)
    xc = (n + chunk - 1) // chunk
    per = (xc + team - 1) // team
    ranges = []
    for r in range(team):
        lo = min(r * per, xc)
        hi = min(lo + per, xc)
        ranges.append((lo * chunk, min(hi * chunk, n)))
    # pass 1: private rows
    cnt_r = [[0] * bins for _ in range(team)]
    first_r = [[None] * bins for _ in range(team)]
    for r, (lo, hi) in enumerate(ranges):
        for p in range(lo, hi):
            c = capacities[p]
            if c >= bins:
                return None  # (the maximum over the team's words: every wavefront takes the walk)
            if c > 0:
                cnt_r[r][c] += 1
                if first_r[r][c] is None:
                    first_r[r][c] = p
    # behind the barrier: everybody reads everybody's rows
    cnt = [sum(cnt_r[r][c] for r in range(team)) for c in range(bins)]
    first = [min((first_r[r][c] for r in range(team) if first_r[r][c] is not None), default=None) for c in range(bins)]
    pre_r = [[sum(cnt_r[q][c] for q in range(r)) for c in range(bins)] for r in range(team)]
    S = sum(cnt[c] * min(c, K) for c in range(bins))
    if S < K:
        return None, False
    max_cap = max(c for c in range(bins) if cnt[c])
    top = bins
    if K < max_cap:
        target = (K + max_cap) // 2
        if sum(cnt[c] * min(c, K) for c in range(target)) >= K:
            top = target

    def smallest_at_least(need, below):
        return next((c for c in range(max(need, 1), below) if cnt[c]), None)

    def largest_below(below):
        return next((c for c in range(below - 1, 0, -1) if cnt[c]), 0)

    R = K
    cf = smallest_at_least(R, top)
    if cf is not None:
        return [first[cf]] * K, True  # (wavefront 0 emits)
    take, base = {}, {}
    last_pos, next_level, next_rank = None, None, 0
    while True:
        m = largest_below(top)
        assert 0 < m < R
        q = R // m
        drained = min(cnt[m], q)
        take[m], base[m] = drained, K - R
        R -= drained * m
        if R == 0:
            break
        if drained < cnt[m]:
            cf = smallest_at_least(R, m)
            if cf is not None:
                last_pos = first[cf]
            else:
                next_level, next_rank = m, drained
            break
        top = m
        cf = smallest_at_least(R, top)
        if cf is not None:
            last_pos = first[cf]
            break
    last_from_plan = last_pos is not None
    # pass 2: every quarter on its own, its level counts starting at the prefix of the quarters before; it stops when what the
    # plan wants from it is out (`left`)
    out = [None] * K
    writers = [0] * K  # every output word is written exactly once
    found = None
    for r, (lo, hi) in enumerate(ranges):
        left = sum(min(max(take.get(c, 0) - pre_r[r][c], 0), cnt_r[r][c]) for c in range(1, bins))
        if next_level is not None and pre_r[r][next_level] <= next_rank < pre_r[r][next_level] + cnt_r[r][next_level]:
            left += 1
        seen = {c: pre_r[r][c] for c in range(bins)}
        for p in range(lo, hi):
            if left == 0:
                break
            c = capacities[p]
            if c in take or c == next_level:
                rank = seen[c]
                seen[c] = rank + 1
                if rank < take.get(c, 0):
                    for i in range(c):
                        out[base[c] + rank * c + i] = p
                        writers[base[c] + rank * c + i] += 1
                    left -= 1
                if c == next_level and rank == next_rank:
                    found = p
                    left -= 1
        assert left == 0
    if R > 0:
        node = last_pos if last_from_plan else found  # (wavefront 0 / the quarter that found it emits)
        for i in range(R):
            out[K - R + i] = node
            writers[K - R + i] += 1
    assert all(w == 1 for w in writers)
    return out, True


@pytest.mark.parametrize("seed", range(30))
def test_minfrag_team_is_the_single_wavefront_form(seed):
    rng = np.random.default_rng(9100 + seed)
    n = int(rng.integers(1, 1200))
    hi = int(rng.choice([2, 5, 12, 40, 255, 300]))
    caps = [int(v) for v in rng.integers(0, hi + 1, size=n)]
    if rng.random() < 0.5:  # neighbours with equal capacities, as a priority order has them
        caps = sorted(caps)
    total = sum(min(c, 255) for c in caps)
    for _ in range(25):
        K = int(rng.integers(0, max(2, min(total + 3, 600))))
        for team in (1, 2, 4, 8):
            assert minfrag_team(K, caps, team=team) == minfrag_histogram(K, caps), (seed, K, team)


def test_minfrag_histogram_doc_comment_examples():
    caps = [1, 1, 3, 5, 5, 17]  # a .. f of minimal_fragmentation.go:43-58
    a, b, c, d, e, f = range(6)
    assert minfrag_histogram(11, caps) == ([d] * 5 + [e] * 5 + [a], True)
    assert minfrag_histogram(6, caps) == ([d] * 5 + [a], True)
    assert minfrag_histogram(15, caps) == ([d] * 5 + [e] * 5 + [c] * 3 + [a, b], True)
    assert minfrag_histogram(17, caps) == ([f] * 17, True)
    for k in range(0, 40):
        assert minfrag_histogram(k, caps) == minfrag_reference(k, caps)


@pytest.mark.parametrize("seed", range(60))
def test_minfrag_histogram_is_the_walk_over_the_sorted_list(seed):
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(1, 400))
    hi = int(rng.choice([2, 5, 12, 40, 255]))
    caps = [int(v) for v in rng.integers(0, hi + 1, size=n)]
    total = sum(caps)
    for _ in range(40):
        k = int(rng.integers(0, max(2, total + 3)))
        got = minfrag_histogram(k, caps)
        assert got is not None
        assert got == minfrag_reference(k, caps), (k, caps)
    assert minfrag_histogram(3, caps + [256]) is None  # beyond the last bin: the walk takes over


# ------------------------------------------------------------------------------------------------ snapshot build: the units' gcds
# finalize_slots_kernel: every lane takes its value modulo a candidate (some non-zero value of the chunk); all remainders zero =>
# the candidate, itself a member, is the gcd; else the candidate becomes its gcd with one non-zero remainder (at most half of it).

def chunk_gcd_rounds(values):
    import math
    v = [abs(int(x)) for x in values]
    any_or = 0
    for x in v:
        any_or |= x
    if any_or == 0:
        return 0, 0
    tz = (any_or & -any_or).bit_length() - 1            # the common power of two comes from the OR of the magnitudes
    w = [x >> tz for x in v]
    cand = next(x for x in w if x)
    rounds = 0
    while True:
        rounds += 1
        rem = [x % cand for x in w]
        left = [r for r in rem if r]
        if not left:
            break
        cand = math.gcd(cand, left[0])                   # wave-uniform Euclid on the first non-zero remainder
    return cand << tz, rounds


def gcd_fast(a, b):
    import math
    if a == b or b == 0:
        return a
    if a == 0:
        return b
    sh = ((a | b) & -(a | b)).bit_length() - 1
    a >>= (a & -a).bit_length() - 1
    b >>= (b & -b).bit_length() - 1
    return math.gcd(a, b) << sh                           # (the device runs Euclid on the odd parts, in 32 bits when both fit)


@pytest.mark.parametrize("seed", range(30))
def test_chunk_gcd_by_candidate_rounds(seed):
    import math
    rng = np.random.default_rng(9100 + seed)
    unit = int(rng.choice([1, 100, 250, 1 << 20, 3 << 28, 1000]))
    n = int(rng.integers(1, 65))
    vals = [int(x) * unit * int(rng.choice([1, -1])) for x in rng.integers(0, 5000, size=n)]
    got, rounds = chunk_gcd_rounds(vals)
    assert got == math.gcd(*[abs(x) for x in vals]) if any(vals) else got == 0
    assert rounds <= 33                                   # the candidate at least halves per round
    for _ in range(50):
        a, b = (int(x) * unit for x in rng.integers(0, 1 << 20, size=2))
        assert gcd_fast(a, b) == math.gcd(a, b)


# ------------------------------------------------------------------------------------------------ resumed chains: the overlay's source
# chain_prologue_kernel, delta format: a chunk named by the cumulative mask comes from the LATEST checkpoint at or before `count`
# whose delta mask names it.  Sequential: walk j = count .. 1.  Device: lane l looks at checkpoint hi - l, 64 at a time from
# hi = count downwards; the lowest lane with its bit set is the latest checkpoint.

def checkpoint_walk(delta, count, chunk):
    for j in range(count, 0, -1):
        if delta[j - 1][chunk]:
            return j
    return 0


def checkpoint_lanes(delta, count, chunk):
    hi = count
    while hi >= 1:
        bits = [(hi - lane >= 1) and bool(delta[hi - lane - 1][chunk]) for lane in range(WAVE)]
        if any(bits):
            return hi - bits.index(True)
        hi = hi - 64 if hi > 64 else 0
    return 0


@pytest.mark.parametrize("seed", range(20))
def test_checkpoint_lookup_by_lanes(seed):
    rng = np.random.default_rng(4400 + seed)
    count = int(rng.choice([1, 2, 31, 63, 64, 65, 127, 200]))
    chunks = 40
    delta = rng.random((count, chunks)) < float(rng.choice([0.02, 0.2, 0.7]))
    for c in range(chunks):
        for upto in {count, max(1, count // 2), 1}:
            assert checkpoint_lanes(delta, upto, c) == checkpoint_walk(delta, upto, c)


# ------------------------------------------------------------------------------------------------ the C oracle against the same walk
# oracle/gangfit_oracle.c is the checker of every GPU parity test; here ITS minimal-fragmentation is held to the Python
# transcription of minimal_fragmentation.go:59-137 above (minfrag_reference) — an independent restatement written from the Go
# source in a different language and shape (lists and slices instead of index arithmetic).

@pytest.mark.parametrize("seed", range(25))
def test_the_c_oracle_agrees_with_the_python_transcription_of_the_go_walk(seed):
    from oracle import binding as ob
    rng = np.random.default_rng(5200 + seed)
    n = int(rng.integers(1, 120))
    hi = int(rng.choice([2, 6, 20, 300]))
    caps = [int(v) for v in rng.integers(0, hi + 1, size=n)]
    # node i takes caps[i] executors of (1 cpu, 1 B); one more node, outside the executor order, hosts the driver
    avail = [[c, 10 ** 6, 0] for c in caps] + [[1, 1, 0]]
    order = [int(v) for v in rng.permutation(n)]
    in_order = [caps[p] for p in order]
    total = sum(caps)
    for _ in range(25):
        k = int(rng.integers(0, max(2, total + 3)))
        ok, drv_node, ex = ob.spark_binpack(ob.ALGO_MINIMAL_FRAGMENTATION, avail, [1, 1, 0], [1, 1, 0], k, [n], order)
        want, want_ok = minfrag_reference(k, in_order)
        assert ok == want_ok, (k, caps)
        if ok:
            assert drv_node == n
            assert [int(v) for v in ex] == [order[p] for p in want], (k, in_order)
