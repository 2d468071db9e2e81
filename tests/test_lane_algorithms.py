"""CPU models of two lane-parallel device routines of the zone-aware chains (csrc/gangfit_fifo_zoned.inc), checked against the
sequential definitions they replace.  The GPU parity tests compare the kernels' float64 results with the oracle bit for bit;
these models document WHY the routines are exact, step by step, and run without a GPU.

  wave_serial_sum_runs   the slice-order sum of ComputeAvgPackingEfficiency (efficiency.go:114-156) as a systolic pass
  zoned_choose_best      chooseBestResult (single_az.go:75-97) as a row max-scan
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
