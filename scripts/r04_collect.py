#!/usr/bin/env python3
"""Copy the round-4 evidence that scripts/r04_profiles.sh left under gpurun_out/prof/r04/ into profiles/ (tracked).

The PMC summaries keep the `source_hash` of the kernel sources they were measured on; bench.py reports their traffic only while
that hash is the current one.  Usage: python3 scripts/r04_collect.py [gpurun_out/prof/r04]
"""
import os
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof/r04"
dst = "profiles"
pairs = [
    ("pmc_intersect.json", "r04_pmc_intersect.json"),
    ("pmc_merge.json", "r04_pmc_merge.json"),
    ("pmc_merge_to_segment.json", "r04_pmc_merge_to_segment.json"),
    ("pmc_c5.json", "r04_pmc_c5.json"),
    ("sq_intersect.txt", "r04_sq_intersect.txt"),
    ("sq_merge.txt", "r04_sq_merge.txt"),
    ("sq_encode.txt", "r04_sq_encode.txt"),
    ("sq_c5.txt", "r04_sq_c5.txt"),
    ("bench_kernel_stats.csv", "r04_bench_kernel_stats.csv"),
    ("bench_under_profiler.json", "r04_bench_under_profiler.json"),
    ("source_hash.txt", "r04_source_hash.txt"),
]
for a, b in pairs:
    p = os.path.join(src, a)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copyfile(p, os.path.join(dst, b))
        print("copied", b)
    else:
        print("MISSING", a)
