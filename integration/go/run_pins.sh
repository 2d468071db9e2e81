#!/bin/bash
# Close the parity pin in ONE command on a machine with Go 1.19 and a checkout of palantir/k8s-spark-scheduler (the version
# this repository was built against: go.mod `k8s-spark-scheduler-lib v0.15.0`, vendored):
#
#     integration/go/run_pins.sh /path/to/k8s-spark-scheduler            # diff: the reference's OWN packers vs the fixtures
#     integration/go/run_pins.sh /path/to/k8s-spark-scheduler -update    # rewrite the fixtures' answers from the reference,
#                                                                        # copy them back into tests/golden/
#     integration/go/run_pins.sh /path/to/k8s-spark-scheduler -bench     # also: the true Go CPU baseline (binpack_bench_test.go)
#
# What it does: copies golden_test.go + golden_snapshot_test.go (package extender: they call the unexported
# sparkResourceUsage and findNodes) and tests/golden/gangfit_golden_v{1,2,3,4}.json into <checkout>/internal/extender/, runs
# `go test -mod=vendor -run 'TestGangfitGolden' ./internal/extender/` there, and with -update copies the rewritten JSON files
# back.  After an -update, `python -m pytest tests/test_golden.py tests/test_golden_v4.py` holds the oracle (CPU) and — with
# `-m gpu` — the HIP path to vectors the reference itself produced: the rows DESIGN.md section 5 lists as "parity unpinned"
# (distributeExecutorsEvenly, FIFO replay with earlier drivers, findNodes) are then pinned.  Nothing is left behind in the
# checkout.  The build container of this repository has no Go toolchain: this script is UNVERIFIED there.
set -euo pipefail
REF=${1:?usage: run_pins.sh <reference checkout> [-update] [-bench]}
shift
UPDATE=""; BENCH=0
for a in "$@"; do
  case $a in
    -update) UPDATE="-update" ;;
    -bench) BENCH=1 ;;
    *) echo "unknown argument $a" >&2; exit 2 ;;
  esac
done
HERE=$(cd "$(dirname "$0")" && pwd)
REPO=$(cd "$HERE/../.." && pwd)
command -v go >/dev/null || { echo "go toolchain not found (Go 1.19 expected)" >&2; exit 3; }
[ -f "$REF/internal/extender/resource.go" ] || { echo "$REF is not a k8s-spark-scheduler checkout" >&2; exit 3; }
DST=$REF/internal/extender
FILES="golden_test.go golden_snapshot_test.go"
FIXTURES="gangfit_golden_v1.json gangfit_golden_v2.json gangfit_golden_v3.json gangfit_golden_v4.json"
cleanup() {
  for f in $FILES; do rm -f "$DST/gangfit_$f"; done
  for f in $FIXTURES; do rm -f "$DST/$f"; done
  rm -f "$REF/internal/binpacker/gangfit_binpack_bench_test.go"
}
trap cleanup EXIT
for f in $FILES; do cp "$HERE/$f" "$DST/gangfit_$f"; done          # (*_test.go suffix kept: gangfit_golden_test.go ...)
for f in $FIXTURES; do cp "$REPO/tests/golden/$f" "$DST/$f"; done
rc=0
( cd "$REF" && go test -mod=vendor -count=1 -run 'TestGangfitGolden' ./internal/extender/ $UPDATE ) || rc=$?
if [ -n "$UPDATE" ] && [ $rc -eq 0 ]; then
  for f in $FIXTURES; do cp "$DST/$f" "$REPO/tests/golden/$f"; done
  echo "fixtures rewritten from the reference: now run  python -m pytest tests/test_golden.py tests/test_golden_v4.py  (and -m gpu on the MI355X box)"
fi
if [ $BENCH -eq 1 ]; then
  cp "$HERE/binpack_bench_test.go" "$REF/internal/binpacker/gangfit_binpack_bench_test.go"
  ( cd "$REF" && go test -mod=vendor -run '^$' -bench . -benchtime 3x ./internal/binpacker/ )
fi
exit $rc
