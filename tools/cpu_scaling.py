#!/usr/bin/env python3
"""How the CPU oracle scales with OpenMP threads on THIS host, under different OpenMP environments (the round-4
`cpu_baseline.by_threads` collapsed above 16 threads on the 256-thread GPU box).  Each (environment, thread count) runs in its own
process: libgomp reads its environment once.  Prints one JSON line per run and a summary table.

usage: tools/cpu_scaling.py [--size 4096] [--wname db4] [--levels 3] [--threads 1,16,32,64,128,256] [--seconds 1.5]
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ENVS = {
    "default": {},
    "bind_spread_cores": {"OMP_PROC_BIND": "spread", "OMP_PLACES": "cores"},
    "bind_close_threads": {"OMP_PROC_BIND": "close", "OMP_PLACES": "threads"},
    "passive": {"OMP_WAIT_POLICY": "passive"},
    "bind_spread_passive": {"OMP_PROC_BIND": "spread", "OMP_PLACES": "cores", "OMP_WAIT_POLICY": "passive"},
}


def host_info():
    info = {"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/proc/loadavg"):
        try:
            info[p] = open(p).read().strip()
        except OSError:
            pass
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True).stdout
        for line in out.splitlines():
            k = line.split(":")[0].strip()
            if k in ("Model name", "Socket(s)", "Core(s) per socket", "Thread(s) per core", "NUMA node(s)", "Hypervisor vendor"):
                info[k] = line.split(":", 1)[1].strip()
    except OSError:
        pass
    return info


def child(size, wname, levels, nthreads, seconds):
    sys.path.insert(0, ROOT)
    import numpy as np
    from oracle import oracle as orc
    x = np.random.RandomState(0).uniform(0, 255, (size, size)).astype(np.float32)
    W = orc.OracleWavelets(x, wname, levels)
    used = orc.set_num_threads(nthreads)

    def pair():
        W.forward()
        W.inverse()
    pair()
    t0 = time.perf_counter()
    pair()
    t1 = time.perf_counter() - t0
    reps = int(max(1, min(100, seconds / max(t1, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(reps):
        pair()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"threads": used, "mpix_s": round(size * size / dt / 1e6, 1), "s_per_pair": round(dt, 5), "reps": reps}))


def main():
    a = sys.argv[1:]
    def opt(name, default):
        return a[a.index(name) + 1] if name in a else default
    size, wname, levels = int(opt("--size", 4096)), opt("--wname", "db4"), int(opt("--levels", 3))
    seconds = float(opt("--seconds", 1.5))
    if "--child" in a:
        return child(size, wname, levels, int(opt("--child", 1)), seconds)
    threads = [int(t) for t in opt("--threads", "1,8,16,32,64,128,256").split(",")]
    hi = host_info()
    print(json.dumps({"host": hi}), flush=True)
    threads = sorted({t for t in threads if t <= hi["cpu_count"]})
    table = {}
    for name, env in ENVS.items():
        for t in threads:
            e = dict(os.environ)
            e.update(env)
            e["OMP_NUM_THREADS"] = str(t)
            try:
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(t), "--size", str(size), "--wname", wname, "--levels", str(levels),
                                      "--seconds", str(seconds)], env=e, capture_output=True, text=True, timeout=120)
                r = json.loads(out.stdout.strip().splitlines()[-1])
            except Exception as ex:  # a run that hangs or dies is a data point too
                r = {"threads": t, "mpix_s": None, "error": repr(ex)[:100]}
            r["env"] = name
            table.setdefault(name, {})[t] = r["mpix_s"]
            print(json.dumps(r), flush=True)
    print("\n| env | " + " | ".join(str(t) for t in threads) + " |")
    print("|---|" + "---|" * len(threads))
    for name in ENVS:
        print("| %s | " % name + " | ".join(str(table[name].get(t)) for t in threads) + " |")


if __name__ == "__main__":
    main()
