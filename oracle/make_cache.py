#!/usr/bin/env python
"""Runs the CPU oracle over the parity workloads (oracle/workloads.py) and stores the compact results under
tests/golden/oracle_cache/ (oracle/cache.py).  TEST INFRASTRUCTURE ONLY.

  python oracle/make_cache.py                       # every workload
  python oracle/make_cache.py bench_vit_h_T24 cfg4_pips_3obj

Run in the build container (no GPU needed); the files travel to the GPU box with the working tree, so a `gpurun` call spends
its minutes on the HIP path.  They are git-ignored: a fresh clone (the driver's round-end run) computes the oracle live."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import workloads as W  # noqa: E402
from oracle.cache import path_for  # noqa: E402

ALL = {"bench_vit_b_T8": lambda: W.bench_workload("vit_b", 8), "bench_vit_h_T24": lambda: W.bench_workload("vit_h", 24),
       **{name: (lambda n=name: W.config_workload(n)) for name in W.CONFIGS}}

if __name__ == "__main__":
    names = sys.argv[1:] or list(ALL)
    for name in names:
        w = ALL[name]()
        key = W.key_of(w)
        if os.path.exists(path_for(key)):
            print(f"{name}: cached ({path_for(key)})", flush=True)
            continue
        t0 = time.time()
        ref = W.reference(w)
        print(f"{name}: {time.time() - t0:.0f} s -> {path_for(key)} ({os.path.getsize(path_for(key)) / 1e6:.1f} MB); "
              f"oracle seconds {ref['seconds']}", flush=True)
