cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_pre
BENCH_PRETEND=${1:-3}/8 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_pre -o pre -- python3 bench.py --workload strong --steps 6 --warmup 2 > gpurun_out/pre.json 2> gpurun_out/pre.err
