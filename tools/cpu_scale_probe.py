#!/usr/bin/env python3
"""How the CPU oracle (torch CPU U-Net) scales on this host: workers x threads, two windows of the base-64 model per
worker.  Decided the cpu_baseline configuration of bench.py (one single-threaded worker per usable core, at most 16)."""
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
import multiprocessing as mp
if __name__ == '__main__':
    ctx = mp.get_context('spawn')
    for nproc, threads in ((1, 1), (16, 1), (64, 1), (128, 1), (16, 8), (32, 4)):
        t0 = time.time()
        with ctx.Pool(nproc, initializer=bench._cpu_init, initargs=(ctx.Value('i', 0), threads)) as pool:
            pool.map(bench._cpu_noop, range(nproc))
            t1 = time.time()
            out = pool.map(bench._cpu_worker, [(64, 900 + i, threads, 2) for i in range(nproc)], chunksize=1)
        ts = [o[3] for o in out]
        print('nproc %3d threads %d: spawn %.1f s, U-Net on 2 windows per worker: mean %.2f s max %.2f s -> %.2f images/s equivalent'
              % (nproc, threads, t1 - t0, sum(ts) / len(ts), max(ts), nproc / (max(ts) * 17.5 + 0.6)), flush=True)
