"""Soak: the concurrency test of tests/test_gpu_threads.py (four contexts, kernels that wait between workgroups) N times in
one process, with and without a tracer-like perturbation (random short sleeps on the launching threads)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inverted_index_2_amd import Context
from tests import test_gpu_threads as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = Context(0)
t0 = time.time()
for i in range(n):
    T.test_kernels_that_wait_between_workgroups_do_not_starve_each_other_across_contexts(ctx)
    print("round", i, "ok", round(time.time() - t0, 1), "s", flush=True)
print("counters of the main context:", ctx.counters())
