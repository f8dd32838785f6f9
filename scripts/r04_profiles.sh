#!/bin/bash
# Round-4 evidence in one GPU-box call.  Everything lands under gpurun_out/prof/r04/; the summaries to keep are copied into profiles/
# by hand afterwards (scripts/r04_collect.py).  Separate rocprofv3 runs: --pmc passes on their own, --kernel-trace --stats on its own.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof/r04
mkdir -p $out
hash=$(python3 -c "import bench; print(bench.kernel_sources_hash())")
echo "source_hash $hash" > $out/source_hash.txt
pmc() {      # pmc <name> <first-kernel-of-a-pass> <passes> <script> [args...]
    name=$1; first=$2; passes=$3; shift 3
    mkdir -p $out/pmc_$name
    timeout -k 5 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_$name/fetch -o fetch -- python3 "$@" > $out/pmc_$name/fetch.log 2>&1 || { echo "pmc $name fetch failed"; return 1; }
    timeout -k 5 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_$name/write -o write -- python3 "$@" > $out/pmc_$name/write.log 2>&1 || { echo "pmc $name write failed"; return 1; }
    python3 scripts/pmc_passes.py $out/pmc_$name/fetch/fetch_counter_collection.csv $out/pmc_$name/write/write_counter_collection.csv $out/pmc_$name.json "$first" "$passes" \
        "{\"source_hash\": \"$hash\", \"command\": \"python3 $*\"}" > $out/pmc_$name.summary.txt
    grep -h "device\|algorithmic" $out/pmc_$name/fetch.log | tail -2 >> $out/pmc_$name.summary.txt
    echo "pmc $name done"
}
sq() {       # sq <name> <kernel-substring> <script> [args...]
    name=$1; kern=$2; shift 2
    bash scripts/r4_pmc.sh r04_sq_$name "$kern" "$@" > /dev/null 2>&1 && cp gpurun_out/r04_sq_${name}_pmc.txt $out/sq_$name.txt && echo "sq $name done" || echo "sq $name failed"
}
pmc intersect ii2::k_and2_fused 20 scripts/c2_loop.py steps=20
pmc merge ii2::k_mp_terms 3 scripts/merge_loop.py steps=3
pmc merge_to_segment ii2::k_mp_terms 3 scripts/m2s_loop.py steps=3
pmc c5 ii2::k_isect_partition 10 scripts/c5_loop.py steps=10
sq intersect k_and2 scripts/c2_loop.py steps=5
sq merge k_merge_tiles scripts/merge_loop.py steps=3
sq encode k_enc_stream scripts/m2s_loop.py steps=3
sq c5 k_isect scripts/c5_loop.py steps=5
# the PMC summaries go where bench.py looks for them (profiles/, this box's copy of the tree): the bench lines below then carry `traffic`
python3 scripts/r04_collect.py $out > /dev/null
mkdir -p $out/bench
timeout -k 5 1000 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -o bench -- python3 bench.py --steps 20 --warmup 5 > $out/bench_under_profiler.json 2> $out/bench_under_profiler.err || echo "bench profile failed"
find $out/bench -name "*kernel_stats.csv" -exec cp {} $out/bench_kernel_stats.csv \;
ls $out
