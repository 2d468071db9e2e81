import os, sys, time, signal, faulthandler
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd"), os.path.join(REPO, "tests")]
import stress_lib
faulthandler.enable()
seed = int(sys.argv[1]); budget = float(sys.argv[2])
ctxs = stress_lib.make_contexts()
t_end = time.time() + budget
slow = []
while time.time() < t_end:
    faulthandler.dump_traceback_later(45, exit=True)
    print("seed", seed, flush=True) if seed % 50 == 0 else None
    sys.stderr.write(f"at seed {seed}\n") if False else None
    open("/tmp/cur_seed", "w").write(str(seed))
    t0 = time.time()
    c, bad = stress_lib.one_seed(ctxs, seed)
    faulthandler.cancel_dump_traceback_later()
    dt = time.time() - t0
    if dt > 5: slow.append((seed, dt)); print("slow seed", seed, dt, flush=True)
    if bad:
        print(bad); sys.exit(1)
    seed += 1
print("ok up to", seed - 1, "slow", slow)
