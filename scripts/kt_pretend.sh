cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_pre3
BENCH_PRETEND=3/8 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_pre3 -o pre3 -- python3 bench.py --workload strong --steps 6 --warmup 2 > gpurun_out/pre3.json 2> gpurun_out/pre3.err
