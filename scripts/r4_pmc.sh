#!/bin/bash
# SQ counter passes over a script: scripts/r4_pmc.sh <tag> <kernel-name-substring> <script> [args...]  -> gpurun_out/<tag>_pmc.txt
tag=$1; kern=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof/$tag
timeout -k 5 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d gpurun_out/prof/$tag/p1 -o p1 -- python3 "$@" > gpurun_out/prof/$tag/p1.log 2>&1 || exit 1
timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/prof/$tag/p2 -o p2 -- python3 "$@" > gpurun_out/prof/$tag/p2.log 2>&1 || exit 1
python3 - "$tag" "$kern" <<'PY' > gpurun_out/${tag}_pmc.txt
import csv, collections, sys
tag, kern = sys.argv[1], sys.argv[2]
for pas in ("p1", "p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open("gpurun_out/prof/%s/%s/%s_counter_collection.csv" % (tag, pas, pas))):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if kern in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k, {c: round(sum(v[1:]) / max(len(v[1:]), 1)) for c, v in d.items()})
PY
cat gpurun_out/${tag}_pmc.txt
