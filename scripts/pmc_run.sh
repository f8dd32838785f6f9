#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (separate runs, --pmc only) for one workload script; summary -> gpurun_out/prof/<tag>/pmc.json
# usage: scripts/pmc_run.sh <tag> <first-kernel-of-a-pass> <passes> <script.py> [args...]
tag=$1; first=$2; passes=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof/$tag
timeout -k 5 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/$tag/fetch -o fetch -- python3 "$@" > gpurun_out/prof/$tag/fetch.log 2>&1 || exit 1
echo fetch pass done
timeout -k 5 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/$tag/write -o write -- python3 "$@" > gpurun_out/prof/$tag/write.log 2>&1 || exit 1
echo write pass done
python3 scripts/pmc_passes.py gpurun_out/prof/$tag/fetch/fetch_counter_collection.csv gpurun_out/prof/$tag/write/write_counter_collection.csv gpurun_out/prof/$tag/pmc.json "$first" "$passes"
grep -h "device\|algorithmic" gpurun_out/prof/$tag/fetch.log | tail -2
