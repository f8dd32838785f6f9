#!/bin/bash
# kernel-trace stats of an arbitrary python command.  Usage: scripts/prof_cmd.sh <tag> <script> [args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof/$tag
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/$tag/kt -o kt -- python3 "$@" > gpurun_out/prof/$tag/kt.log 2>&1 || { tail -5 gpurun_out/prof/$tag/kt.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof/$tag/kt/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:16]:
    print("%-62s calls %6s avg_us %10.1f total_ms %9.2f  %5s%%" % (r["Name"].split("(")[0][:62], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
