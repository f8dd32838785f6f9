#!/bin/bash
# kernel-trace of a probe script: per-kernel average durations -> gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 "$@" > gpurun_out/${tag}_prof.log 2>&1
find gpurun_out/prof_$tag -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_kernel_stats.csv \;
