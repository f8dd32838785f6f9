#!/bin/bash
# merge_strong: chunks x workers, whole job on one GPU and one rank's share (BENCH_PRETEND=r/8)
for c in 2 3 4; do for w in 2 3; do
  python bench.py --workload strong --steps 10 --warmup 3 --strong-chunks $c --strong-workers $w 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=1 chunks $c workers $w', round(d['ms_per_step'],2), round(d['merge_to_segment_only']['ms_per_step'],2))"
  for r in 0 5; do BENCH_PRETEND=$r/8 python bench.py --workload strong --steps 30 --warmup 5 --strong-chunks $c --strong-workers $w 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   rank $r chunks $c workers $w', round(d['ms_per_step'],2))"; done
done; done
