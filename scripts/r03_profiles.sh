#!/bin/bash
# Round-3 evidence, one GPU box call: HBM traffic of the merge and of the intersection (FETCH_SIZE / WRITE_SIZE passes), the SQ
# counters of the merge kernels, kernel-trace stats of the default bench.  Everything lands under gpurun_out/prof/r03/.
cd $GRAFT_REPO_ROOT
scripts/pmc_run.sh r03/pmc_merge ii2::k_mp_terms 3 scripts/merge_loop.py steps=3 || echo "pmc merge failed"
scripts/pmc_run.sh r03/pmc_isect ii2::k_dense_tiles 20 scripts/c2_loop.py steps=20 || echo "pmc isect failed"
scripts/pmc_merge_sq.sh r03/sq_merge terms=1000000 > gpurun_out/prof/r03/sq_merge.log 2>&1 || echo "sq merge failed"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof/r03/bench
timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/r03/bench -o bench -- python3 bench.py --steps 20 --warmup 5 > gpurun_out/prof/r03/bench.json 2> gpurun_out/prof/r03/bench.err || echo "bench profile failed"
ls gpurun_out/prof/r03 gpurun_out/prof/r03/*
