cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_soak
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_soak -o soak -- python3 scripts/soak_contexts.py 3 > gpurun_out/soak_kt.log 2>&1
tail -3 gpurun_out/soak_kt.log
II2_OPTIONS=debug.no_chain=1 timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_soak2 -o soak2 -- python3 scripts/soak_contexts.py 3 > gpurun_out/soak_kt2.log 2>&1
tail -4 gpurun_out/soak_kt2.log
rm -rf gpurun_out/prof_soak gpurun_out/prof_soak2
