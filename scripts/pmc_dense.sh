#!/bin/bash
# PMC + kernel-trace passes over the dense intersection kernels (C2).  Usage: scripts/pmc_dense.sh <tag> [c2_loop args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof/$tag
timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/$tag/kt -o kt -- python3 scripts/c2_loop.py steps=30 "$@" > gpurun_out/prof/$tag/kt.log 2>&1 || exit 1
timeout -k 5 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d gpurun_out/prof/$tag/p1 -o p1 -- python3 scripts/c2_loop.py steps=5 "$@" > gpurun_out/prof/$tag/p1.log 2>&1 || exit 1
timeout -k 5 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/prof/$tag/p2 -o p2 -- python3 scripts/c2_loop.py steps=5 "$@" > gpurun_out/prof/$tag/p2.log 2>&1 || exit 1
python3 - <<PY
import csv, collections
for f in ("kt/kt_kernel_stats.csv",):
    for r in list(csv.DictReader(open("gpurun_out/prof/$tag/"+f)))[:3]:
        print(r["Name"][:60], r["Calls"], r["AverageNs"])
for pas in ("p1", "p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open("gpurun_out/prof/$tag/%s/%s_counter_collection.csv" % (pas, pas))):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "dense" in k or "and2" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k, {c: round(sum(v[1:]) / max(len(v[1:]), 1)) for c, v in d.items()})
PY
