#!/usr/bin/env python3
"""bench.py — posting-list hot path on MI355X: postings/s and fraction of the HBM roofline.

One run times BOTH halves of the metric ("intersect + segment-merge") and prints ONE JSON line on rank 0:

* headline (`value`, `roofline`, `cpu_baseline`): BASELINE.json configs[1] — 2-term intersection over a 100M-doc
  index, Zipf ranks 2 and 3 (≈50M and ≈33M postings), block-Δ-varint (DV1) lists resident in HBM, decoded in-kernel;
  output = the ascending doc ids of the intersection, resident in HBM.  A step = one pass over that input.
  `roofline.cold_*`: the same passes rotated over 4 distinct list pairs (4 x 150 MB of traffic > the 256 MiB Infinity
  Cache), so the figure cannot be fed by the cache.
* `merge` (N = 1): configs[2] — 16-way segment merge of 1M terms x mean 1000 postings (~1.06e9 postings in), 1 %
  tombstones; checked against the oracle before timing; that oracle run is also the `cpu_baseline` sample (the whole
  workload with >= 64 host cores; with fewer, the tail of the term range at ~3M postings per core, the head terms
  checked for order and tombstones only: --cpu-sample).
* `merge_strong` (every N): configs[3] — ONE fixed problem, 64 segments x 1M terms, the terms cut into N contiguous
  ranges balanced by estimated merge cost (shard.go:362-378 ranges are contiguous too); every rank merges its range in
  chunks into DV1 segments (what Shard.Merge writes) and the merged segments are exchanged ENCODED with ii2_seg_allgather
  (RCCL), chunk i on a second context while chunk i + 1 merges.  Its `value` counts the exchange: this is the
  STRONG-scaling figure the 1 -> 8 GPU target is about.  (The headline is WEAK scaling: every rank intersects its own
  100M-doc universe, no exchange in the timed region.)
* `c5` (every N): configs[4] — one 8-term AND over a 1B-doc index (Zipf ranks 2 ... 16384, a common core in every list),
  the doc-id space cut into N contiguous ranges (sharding.doc_range), every rank checked bit-exactly against the oracle on
  its range, results concatenated in rank order with ii2_allgatherv inside the timed region.

N > 1: one process per GPU.  `python bench.py --gpus N` starts the N ranks itself (torch.distributed.run on 127.0.0.1)
when no launcher did (WORLD_SIZE unset) — before this process touches the GPU; under a launcher the ranks read
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*.  Ranks do not talk during the timed steps of the intersection (the
reference's shards are independent, inverted_index.go:83-103).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
METRIC = "postings/sec (intersect + segment-merge) at 1/2/4/8 MI355X; % HBM roofline"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--docs", type=int, default=100_000_000, help="doc-id universe per GPU (config 2: 100M)")
    ap.add_argument("--workload", choices=["all", "intersect", "merge", "strong", "c5"], default="all")
    ap.add_argument("--tombstones", action="store_true", help="apply a 1%% tombstone bitmap during the intersection")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather-timeout", type=int, default=120, help="seconds the all-gatherv exchange may take (N > 1)")
    ap.add_argument("--merge-terms", type=int, default=1_000_000, help="merge workloads: aligned terms")
    ap.add_argument("--merge-segments", type=int, default=16)
    ap.add_argument("--strong-segments", type=int, default=64)
    ap.add_argument("--merge-mean", type=float, default=1000.0)
    ap.add_argument("--merge-steps", type=int, default=0, help="timed merges (0: min(steps, 10))")
    ap.add_argument("--cold-pairs", type=int, default=4)
    ap.add_argument("--strong-chunks", type=int, default=3, help="merge_strong: chunks a rank's term range is merged and exchanged in")
    ap.add_argument("--strong-last-share", type=float, default=0.5,
                    help="merge_strong: cost share of a rank's LAST chunk relative to the others (its exchange is the only one no later "
                         "merge hides)")
    ap.add_argument("--strong-workers", type=int, default=2,
                    help="merge_strong: chunks merged side by side, one context (stream + scratch) each - the reference's "
                         "InvertedIndex.Merge(…, concurrency) fan-out (inverted_index.go:62-109) on one GPU")
    ap.add_argument("--c5-docs", type=int, default=1_000_000_000, help="c5: doc-id universe of the whole index (config 5: 1B)")
    ap.add_argument("--c5-steps", type=int, default=0, help="c5: timed queries (0: min(steps, 20))")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: the ranks only come up (gloo), count themselves and rank 0 prints the line's frame - checks the launch")
    ap.add_argument("--cpu-sample", choices=["auto", "full", "bounded"], default="auto",
                    help="merge: what the oracle checks and times - the whole workload (auto with >= 64 host cores) or the tail of "
                         "the term range, about 3M postings per core (auto with fewer)")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 on a ONE-GPU box: every rank uses cuda:0, the process group is gloo and the exchange takes the "
                         "torch fallback (RCCL refuses two ranks on one device) - checks the multi-rank control flow, the numbers mean nothing")
    return ap.parse_args()


def self_launch(args):
    """`bench.py --gpus N` without a launcher: the N ranks are started here, as children, before this process touches
    the GPU (a process that has initialised the GPU must not be replaced or forked).  Rank 0's JSON line passes through
    on stdout; the exit code is the launcher's (non-zero when any rank failed)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL between processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, cores_available() // max(args.gpus, 1))))
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """The launch, without a GPU: every rank joins a gloo group, the ranks count themselves, rank 0 prints the frame."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    n = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        n = int(t.item())
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "postings/s", "n_gpus": world, "ranks_counted": n, "steps": args.steps,
                          "warmup": args.warmup, "dry_run": True}), flush=True)
    return 0 if n == world else 6


def cores_available():
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def kernel_sources_hash():
    """sha256 over the library's kernel and host sources (csrc/): what a PMC summary under profiles/ was measured on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "inverted_index_2_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".cpp", ".h")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def pmc_traffic(name):
    """HBM bytes per pass from the PMC passes kept under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over the same
    workload, scripts/r04_profiles.sh) — a recorded figure, not a counter of this run; the JSON says so (`traffic_source`).
    Every summary carries the hash of the sources it was measured on (`source_hash`): when the sources have changed since,
    the figure is stale and the line reports null (with the reason) instead of a number that no longer describes the kernels."""
    path = os.path.join(ROOT, "profiles", "r04_pmc_%s.json" % name)
    try:
        with open(path) as f:
            d = json.load(f)
        rel = os.path.relpath(path, ROOT)
        if d.get("source_hash") != kernel_sources_hash():
            return None, "%s is stale: taken on sources %s, this tree is %s" % (rel, d.get("source_hash"), kernel_sources_hash())
        return d["hbm_bytes_per_pass_corrected"], rel
    except (OSError, KeyError, ValueError):
        return None, None


def cpu_baseline_intersect(lists, removed, reps, threads=1):
    """The oracle (CPU restatement: DV1 decode + two-pointer intersection).  threads > 1: the doc range is cut into
    `threads` shards (as the multi-GPU path does) and a worker pool runs one shard each — ctypes releases the GIL."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    n_post = int(sum(l.size for l in lists))
    hi = int(max(int(l[-1]) for l in lists if l.size)) + 1 if n_post else 1
    cuts = [hi * i // threads for i in range(threads + 1)]
    shards = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        part = [l[np.searchsorted(l, a):np.searchsorted(l, b)] for l in lists]
        po = np.concatenate([[0], np.cumsum([x.size for x in part])]).astype(np.uint64)
        flat = np.concatenate(part) if part else np.empty(0, np.uint32)
        shards.append((orc.dv1_encode(po, flat), int(flat.size), len(part)))
    rm = removed if removed is not None else ()

    def work(sh):
        (blk, skip, payload), n, k = sh
        po2, vals = orc.dv1_decode(blk, skip, payload, n)
        return orc.intersect([vals[int(po2[i]):int(po2[i + 1])] for i in range(k)], rm)

    res = None
    with ThreadPoolExecutor(max_workers=threads) as pool:
        t0 = time.perf_counter()
        for _ in range(reps):
            res = list(pool.map(work, shards))
        dt = (time.perf_counter() - t0) / reps
    return n_post / dt, np.concatenate(res) if res else np.empty(0, np.uint32), dt


class Job:
    """Process-wide state of one bench run."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.args, self.torch, self.dist = args, torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
        self.rehearse = bool(args.rehearse) and self.world > 1
        device = 0 if self.rehearse else self.local_rank
        self.dev = "cpu" if self.rehearse else "cuda"      # where the control-plane tensors of the collectives live
        torch.cuda.set_device(device)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.rehearse:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        from inverted_index_2_amd import Context
        self.ctx = Context(device)
        self.ctx.selftest()
        self.parts = {}            # the bench objects measured so far (what the watchdog prints if a later exchange gets stuck)
        if self.world > 1 and self.rank == 0:
            import signal

            def on_term(signum, frame):      # the launcher ends the job because another rank failed: leave what has been measured
                if self.parts:
                    try:
                        line = assemble(self, dict(self.parts))
                        line["aborted"] = "terminated by the launcher (another rank failed); objects measured after that are missing"
                        print(json.dumps(line), flush=True)
                    except Exception:  # noqa: BLE001
                        pass
                os._exit(3)
            try:
                signal.signal(signal.SIGTERM, on_term)
            except (ValueError, OSError):
                pass
        self.comm = None           # "ii2" | "torch" once decided (identically on every rank)
        self.comm_main = False     # the job's own context has its communicator
        self.rc = 0

    def sync_all(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def init_comm(self, ctx=None):
        """The library's own RCCL communicator (torch's only carries the unique id) for `ctx` (default: the job's context).
        Every rank learns whether EVERY rank got one — ranks must never run different collectives against each other."""
        if self.world == 1:
            return
        if ctx is None:
            if self.comm_main:
                return
            ctx = self.ctx
            self.comm_main = True
        if self.comm == "torch":
            return                    # a communicator already failed somewhere: every rank stays on the torch exchange
        from inverted_index_2_amd import comm_unique_id
        ok = 1
        try:
            if self.rehearse:
                raise RuntimeError("rehearsal on one GPU: the library communicator is not attempted")
            uid = [comm_unique_id() if self.rank == 0 else None]
            self.dist.broadcast_object_list(uid, src=0)
            ctx.comm_init(self.world, self.rank, uid[0])
        except Exception as e:  # noqa: BLE001
            print("rank %d: ii2_comm_init failed: %r" % (self.rank, e), file=sys.stderr, flush=True)
            ok = 0
        t = self.torch.tensor([ok], dtype=self.torch.int32, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        self.comm = "ii2" if int(t.item()) == 1 else "torch"

    def gather(self, local, n_local, cap_per_rank):
        """Rank-order concatenation of the ranks' device arrays (inverted_index.go:330-339).  Returns (DeviceArray,
        counts, implementation)."""
        self.init_comm()
        torch, dist = self.torch, self.dist
        if self.comm == "ii2":
            out = self.ctx.empty(cap_per_rank * self.world + 8)
            counts = self.ctx.allgatherv(local, n_local, out, self.world)
            return out, counts, "ii2_allgatherv (ncclAllGather of counts + grouped ncclSend/ncclRecv over xGMI)"
        # every rank agreed that the library communicator is unusable: padded all_gather through torch's RCCL communicator
        cnt = torch.tensor([n_local], dtype=torch.int64, device=self.dev)
        cnts = [torch.zeros_like(cnt) for _ in range(self.world)]
        dist.all_gather(cnts, cnt)
        host = local.download(n_local)
        mine = torch.from_numpy(np.concatenate([host, np.zeros(cap_per_rank - n_local, np.uint32)]).view(np.int32)).to(self.dev)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine)
        counts = [int(c.item()) for c in cnts]
        packed = np.concatenate([pp[:c].cpu().numpy().view(np.uint32) for pp, c in zip(parts, counts)])
        out = self.ctx.empty(packed.size + 8)
        out.upload(packed)
        return out, counts, "torch.distributed.all_gather fallback (ii2_comm_init failed on some rank)"

    def guarded(self, what, fn):
        """Runs an exchange step under a watchdog: a stuck RCCL exchange is a GPU hang — report it and leave NON-zero."""
        done = threading.Event()

        def on_timeout():
            if not done.is_set():
                print("rank %d: %s did not finish within %d s — aborting" % (self.rank, what, self.args.gather_timeout),
                      file=sys.stderr, flush=True)
                # what has been measured so far is not lost with the stuck step: rank 0 still prints its line (marked), then
                # the process leaves NON-zero - the first N > 1 run is the first time the RCCL exchanges execute at all
                if self.rank == 0 and self.parts:
                    try:
                        line = assemble(self, dict(self.parts))
                        line["aborted"] = "%s did not finish within %d s; objects measured after it are missing" % (what, self.args.gather_timeout)
                        print(json.dumps(line), flush=True)
                    except Exception as e:  # noqa: BLE001
                        print("rank 0: could not assemble the partial line: %r" % (e,), file=sys.stderr, flush=True)
                os._exit(3)

        timer = threading.Timer(self.args.gather_timeout, on_timeout)
        timer.daemon = True
        timer.start()
        try:
            return fn()
        finally:
            done.set()
            timer.cancel()


def bench_intersect(job):
    """Headline: configs[1].  Returns the dict of headline fields."""
    args, ctx, world, rank = job.args, job.ctx, job.world, job.rank
    from inverted_index_2_amd import synth
    D = args.docs
    offset = rank * D
    if offset + D > (1 << 32):
        raise SystemExit("doc-id shards exceed the uint32 id space")
    a = synth.zipf_list(2, D, offset)
    b = synth.zipf_list(3, D, offset)
    removed = tomb = None
    if args.tombstones:
        removed = (synth.geometric_postings(0.01, D, synth.term_seed(10**6), offset)).astype(np.uint32)
        tomb = ctx.tombstones(removed)
    seg = ctx.encode_lists([a, b])
    lists = [(seg, 0), (seg, 1)]
    n_in = int(a.size + b.size)
    cap = min(a.size, b.size) + 512
    out = ctx.empty(cap)
    d_count = ctx.empty(8, np.uint64)

    # correctness of this rank's result before timing: bit-exact id sequence
    _, n_out = ctx.intersect(lists, tomb=tomb, out=out)
    got = out.download(n_out)
    want_np = np.intersect1d(a, b, assume_unique=True)
    if removed is not None:
        want_np = np.setdiff1d(want_np, removed, assume_unique=True)
    if n_out != want_np.size or not np.array_equal(got, want_np):
        raise SystemExit(f"rank {rank}: GPU intersection differs from the numpy cross-check")

    def timed(pairs, steps):
        for i in range(args.warmup):
            ctx.intersect_async(pairs[i % len(pairs)], tomb, out, d_count)
        job.sync_all()
        ctx.profile_region(True)
        t0 = time.perf_counter()
        for i in range(steps):
            ctx.intersect_async(pairs[i % len(pairs)], tomb, out, d_count)
        ctx.profile_region(False)
        job.sync_all()
        dt = time.perf_counter() - t0
        return job.max_over_ranks(dt), ctx.profile_region_ms() * 1e-3

    dt, dev_s = timed([lists], args.steps)
    # per-pass device time sampled with an event pair around single passes (a pair idles the stream ~10 us, so
    # this loop is separate from the timed one)
    ctx.set_option("profile.events", 1)
    ctx.profile_read()
    for _ in range(max(10, min(args.steps, 40))):
        ctx.intersect_async(lists, tomb, out, d_count)
    pass_ms, pass_n = ctx.profile_read()
    ctx.set_option("profile.events", 0)

    info = seg.info
    # algorithmic bytes of one pass (SURVEY.md §8 d): encoded payload + 8 B per block of skip table
    # + 4 B per result id (+ D/8 tombstone bitmap)
    alg_bytes = info.n_bytes + 8 * info.n_blocks + 4 * n_out + (D // 8 if tomb is not None else 0)
    kern_avg_s = dev_s / args.steps
    achieved = alg_bytes / kern_avg_s / 1e9
    traffic, traffic_src = (None, None)
    if D == 100_000_000 and tomb is None:
        traffic, traffic_src = pmc_traffic("intersect")

    # cold variant: rotate over distinct list pairs so that no pass finds its input in the Infinity Cache
    cold = None
    if args.cold_pairs > 1:
        pairs = [lists]
        keep = [seg]
        n_in_cold, alg_cold = n_in, alg_bytes
        for i in range(1, args.cold_pairs):
            a2 = synth.zipf_list(2, D, offset, synth.GLOBAL_SEED + 7919 * i)
            b2 = synth.zipf_list(3, D, offset, synth.GLOBAL_SEED + 7919 * i)
            s2 = ctx.encode_lists([a2, b2])
            keep.append(s2)
            pairs.append([(s2, 0), (s2, 1)])
            _, n2 = ctx.intersect(pairs[-1], tomb=tomb, out=out)
            w2 = np.intersect1d(a2, b2, assume_unique=True)
            if removed is not None:
                w2 = np.setdiff1d(w2, removed, assume_unique=True)
            if n2 != w2.size or not np.array_equal(out.download(n2), w2):
                raise SystemExit(f"rank {rank}: cold pair {i} differs from the numpy cross-check")
            n_in_cold += int(a2.size + b2.size)
            alg_cold += s2.info.n_bytes + 8 * s2.info.n_blocks + 4 * n2 + (D // 8 if tomb is not None else 0)
        steps_c = max(args.cold_pairs, (args.steps // args.cold_pairs) * args.cold_pairs)
        cdt, cdev = timed(pairs, steps_c)
        per_pass_alg = alg_cold / args.cold_pairs
        cold = {"pairs": args.cold_pairs, "steps": steps_c, "ms_per_step": cdt / steps_c * 1e3,
                "value": n_in_cold / args.cold_pairs * world * steps_c / cdt,
                "kernel_avg_us": cdev / steps_c * 1e6, "achieved": per_pass_alg / (cdev / steps_c) / 1e9,
                "frac": per_pass_alg / (cdev / steps_c) / 1e9 / HBM_PEAK_GBS,
                "bytes_touched_between_reuses_MB": round((args.cold_pairs - 1) * per_pass_alg / 1e6, 1)}
        for s2 in keep[1:]:
            s2.free()
        ctx.intersect(lists, tomb=tomb, out=out)       # leave the headline pair's result in `out`

    res = {
        "value": n_in * world * args.steps / dt,
        "ms_per_step": dt / args.steps * 1e3,
        "config": {
            "workload": "2-term intersection (Zipf ranks 2 and 3), %d-doc universe per GPU, DV1 block-delta-varint "
                        "decoded in-kernel, doc-range sharded" % D,
            "postings_per_gpu": n_in, "result_ids_per_gpu": n_out, "tombstones": bool(args.tombstones),
            "encoded_bytes_per_gpu": int(info.n_bytes), "blocks_per_gpu": int(info.n_blocks),
            "parallelism": "docrange%d" % world,
        },
        "roofline": {
            "bound": "hbm", "kernel": "one pass of ii2_intersect: every kernel between the first launch and the last "
                                      "(names in profiles/)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": int(alg_bytes), "kernel_avg_us": kern_avg_s * 1e6, "launches_timed": int(args.steps),
            "timing": "one HIP event pair on the library's stream around the %d timed passes" % args.steps,
            "sampled_pass_us": (pass_ms / pass_n * 1e3) if pass_n else None, "sampled_passes": int(pass_n),
        },
    }
    if cold is not None:
        res["roofline"]["cold_frac"] = cold["frac"]
        res["roofline"]["cold"] = cold
    if tomb is None:
        # north_star's form of the query: the same passes with the tombstone filter on (1 % of the docs removed, a D/8-byte
        # bitmap: +12.5 MB algorithmic at 100M docs) - SURVEY §0 D4: off = the reference's Read, on = north_star
        removed_t = synth.geometric_postings(0.01, D, synth.term_seed(10**6), offset).astype(np.uint32)
        tomb_t = ctx.tombstones(removed_t)
        _, n_t = ctx.intersect(lists, tomb=tomb_t, out=out)
        if not np.array_equal(out.download(n_t), np.setdiff1d(want_np, removed_t, assume_unique=True)):
            raise SystemExit(f"rank {rank}: GPU intersection with tombstones differs from the numpy cross-check")
        for i in range(args.warmup):
            ctx.intersect_async(lists, tomb_t, out, d_count)
        job.sync_all()
        ctx.profile_region(True)
        for i in range(args.steps):
            ctx.intersect_async(lists, tomb_t, out, d_count)
        ctx.profile_region(False)
        job.sync_all()
        t_s = ctx.profile_region_ms() * 1e-3 / args.steps
        alg_t = info.n_bytes + 8 * info.n_blocks + 4 * n_t + D // 8
        res["roofline"]["with_tombstones"] = {"removed_ids": int(removed_t.size), "result_ids": int(n_t), "algorithmic_bytes_per_launch": int(alg_t),
                                              "kernel_avg_us": t_s * 1e6, "achieved": alg_t / t_s / 1e9, "frac": alg_t / t_s / 1e9 / HBM_PEAK_GBS,
                                              "value": n_in * world / t_s}
        tomb_t.free()
        ctx.intersect(lists, tomb=None, out=out)       # leave the headline result in `out`

    # the same two lists as a UNION (PrefixSearch's append + sort + compact over the lists of the matching terms,
    # inverted_index.go:274-292): ii2_union's streaming path (the dense kernel with OR semantics), numpy-checked
    if tomb is None and world == 1:
        try:
            u_out = ctx.empty(int(a.size + b.size) + 16)
            _, n_u = ctx.union(lists, out=u_out)
            ok_u = n_u == np.union1d(a, b).size and bool(np.array_equal(u_out.download(min(n_u, 1 << 22)), np.union1d(a, b)[: 1 << 22]))
            ctx.set_option("profile.events", 1)
            ctx.profile_read()
            u_steps = max(5, min(args.steps, 20))
            t0u = time.perf_counter()
            for _ in range(u_steps):
                ctx.union(lists, out=u_out)
            u_wall = (time.perf_counter() - t0u) / u_steps
            u_ms, u_n = ctx.profile_read()
            ctx.set_option("profile.events", 0)
            u_dev = u_ms / max(u_n, 1) * 1e-3
            alg_u = info.n_bytes + 8 * info.n_blocks + 4 * n_u
            res["union"] = {"value": n_in / u_dev, "unit": "postings/s", "result_ids": int(n_u), "kernel_avg_us": u_dev * 1e6,
                            "wall_us_per_call": u_wall * 1e6, "calls_timed": int(u_n),
                            "roofline": {"bound": "hbm", "achieved": alg_u / u_dev / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": alg_u / u_dev / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": int(alg_u), "traffic": None},
                            "what": "ii2_union of the headline's two lists (synchronous call: it returns the count, one stream wait); device "
                                    "time from an event pair per call, wall time beside it",
                            "check": "equals numpy's union (count; first 4M ids compared)" if ok_u else "MISMATCH"}
            if not ok_u:
                job.rc = 5
            u_out.free()
        except Exception as e:  # noqa: BLE001 — the headline stands on its own
            res["union"] = {"error": repr(e)}

    # the exchange step, once, after the timed region: rank-order concatenation of the results
    job.parts["intersect"] = res           # (measured: a stuck exchange below no longer takes the headline with it)
    if world > 1:
        def exchange():
            job.torch.cuda.synchronize()
            g0 = time.perf_counter()
            gout, counts, impl = job.gather(out, n_out, cap)
            ms = (time.perf_counter() - g0) * 1e3
            total = sum(counts)
            allv = gout.download(total)
            good = bool(np.all(np.diff(allv.astype(np.int64)) > 0) and
                        np.array_equal(allv[sum(counts[:rank]):sum(counts[:rank + 1])], got))
            gout.free()
            return ms, total, impl, good
        ms, total, impl, good = job.guarded("the intersection's all-gatherv", exchange)
        res["allgatherv"] = {"ms": ms, "ids": int(total), "impl": impl,
                             "check": "rank-order concatenation verified on rank 0" if good else "MISMATCH on rank %d" % rank}
        if not good:
            print(f"rank {rank}: all-gatherv result is not the rank-order concatenation", file=sys.stderr, flush=True)
            job.rc = 4

    if not args.no_cpu_baseline and world == 1:       # the CPU baseline is a rank-0, N = 1 figure
        reps = 10                                       # ~4 s single-thread + a few seconds on the pool
        v1, res1, per1 = cpu_baseline_intersect([a, b], removed, reps, threads=1)
        if not np.array_equal(res1, got):
            raise SystemExit("GPU result differs from the oracle")
        avail = cores_available()
        vN, resN, perN = cpu_baseline_intersect([a, b], removed, max(reps, int(8.0 / max(per1 / avail, 1e-3) / 4)), threads=avail)
        if not np.array_equal(resN, got):
            raise SystemExit("GPU result differs from the oracle (sharded run)")
        res["cpu_baseline"] = {
            "value": vN, "unit": "postings/s", "cores": avail, "cores_available": avail, "kind": "port", "value_1_thread": v1,
            "sample": "the full rank-0 workload (DV1 decode + two-pointer intersection of %d postings) cut into %d doc-range "
                      "shards on a %d-thread pool (every core of the affinity mask), %.3f s per repetition; single thread: "
                      "%.2f s per repetition, %d repetitions (oracle/ii2_oracle.c)" % (n_in, avail, avail, perN, per1, reps),
        }
    seg.free()
    out.free()
    return res


def merge_alg_bytes(segs, T, k, docs, n_out):
    """SURVEY §8 d: sum over segments (encoded payload + 8 B per block + 4 (T+1) offsets) + D/8 + 4 |out| + 4 (T+1)."""
    enc = sum(s.info.n_bytes for s in segs)
    nblk = sum(s.info.n_blocks for s in segs)
    return enc + 8 * nblk + 4 * k * (T + 1) + docs // 8 + 4 * n_out + 4 * (T + 1)


def time_merges(job, segs, tomb, out_off, out_vals, steps):
    ctx = job.ctx
    ctx.merge(segs, tomb, out_off, out_vals)
    ctx.set_option("profile.events", 1)
    ctx.profile_read()
    job.sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.merge(segs, tomb, out_off, out_vals)
    job.sync_all()
    dt = job.max_over_ranks(time.perf_counter() - t0)
    dev_ms, dev_n = ctx.profile_read()
    ctx.set_option("profile.events", 0)
    return dt, dev_ms, dev_n


def bench_merge(job):
    """configs[2] on one GPU: 16-way merge of `--merge-terms` terms, oracle-checked, oracle-timed."""
    args, ctx = job.args, job.ctx
    from inverted_index_2_amd import synth
    T, k = args.merge_terms, args.merge_segments
    avail = cores_available()
    g0 = time.perf_counter()
    offs, vals, removed = synth.merge_workload_big(T, k, args.merge_mean, args.docs, threads=min(avail, 32))
    gen_s = time.perf_counter() - g0
    segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
    tomb = ctx.tombstones(removed)
    n_in = int(sum(int(o[-1]) for o in offs))
    out_off = ctx.empty(T + 1, np.uint64)
    out_vals = ctx.empty(n_in)
    _, _, st = ctx.merge(segs, tomb, out_off, out_vals)
    cpu = None
    if not args.no_cpu_baseline:
        # the oracle's worker pool over term ranges (mirrors InvertedIndex.Merge(…, concurrency)) on every core of the
        # affinity mask: ONE run, which is both the correctness check and the CPU baseline sample
        from oracle import oracle as orc
        mode = args.cpu_sample if args.cpu_sample != "auto" else ("full" if avail >= 64 else "bounded")
        t_lo = 0
        if mode == "bounded":
            # few host cores: the giant head terms are one core's serial job each (sort after every fold), so the sample is
            # the tail of the term range holding about 3M postings per core; the head is checked by its invariants below
            per_term = np.zeros(T, np.int64)
            for o in offs:
                per_term += np.diff(o.astype(np.int64))
            tail = np.cumsum(per_term[::-1])
            t_lo = int(T - min(int(np.searchsorted(tail, avail * 3_000_000)) + 1, T))
        s_offs = [o[t_lo:] - o[t_lo] for o in offs]
        s_vals = [v[int(o[t_lo]):] for o, v in zip(offs, vals)]
        n_sample = int(sum(int(o[-1]) for o in s_offs))
        t0 = time.perf_counter()
        w_off, w_vals, _ = orc.merge_segments(s_offs, s_vals, removed, threads=avail)
        cdt = time.perf_counter() - t0
        g_off = out_off.download()
        g_vals = out_vals.download(int(g_off[-1]))
        if not (np.array_equal(g_off[t_lo:] - g_off[t_lo], w_off) and np.array_equal(g_vals[int(g_off[t_lo]):], w_vals)):
            raise SystemExit("GPU merge differs from the oracle")
        if t_lo:        # the terms outside the sample: strictly ascending inside every list, nothing tombstoned
            head = g_vals[: int(g_off[t_lo])]
            asc = np.diff(head.astype(np.int64)) > 0
            asc[(g_off[1:t_lo] - 1).astype(np.int64)[g_off[1:t_lo] > 0]] = True
            if not asc.all() or np.isin(head[:: max(1, head.size // 4_000_000)], removed).any():
                raise SystemExit("GPU merge: a list outside the oracle sample is not ascending or holds a tombstoned id")
        del w_off, w_vals, g_off, g_vals, s_offs, s_vals
        what = ("the whole workload once (%d postings in, %.1f s)" % (n_in, cdt)) if t_lo == 0 else \
               ("terms [%d, %d) of the workload (%d of its %d postings in, %.1f s; the head terms are checked for order and "
                "tombstones only)" % (t_lo, T, n_sample, n_in, cdt))
        # one thread, on a bounded sample: the tail of the term range holding about 20M postings (a few seconds)
        per_term1 = np.zeros(T, np.int64)
        for o in offs:
            per_term1 += np.diff(o.astype(np.int64))
        t1_lo = int(T - min(int(np.searchsorted(np.cumsum(per_term1[::-1]), 20_000_000)) + 1, T))
        o1 = [o[t1_lo:] - o[t1_lo] for o in offs]
        v1 = [v[int(o[t1_lo]):] for o, v in zip(offs, vals)]
        n1 = int(sum(int(o[-1]) for o in o1))
        t0 = time.perf_counter()
        orc.merge_segments(o1, v1, removed, threads=1)
        c1 = time.perf_counter() - t0
        del o1, v1
        cpu = {"value": n_sample / cdt, "unit": "postings/s", "cores": avail, "cores_available": avail, "kind": "port",
               "value_1_thread": n1 / c1,
               "sample": what + ": oracle worker pool over term ranges, pairwise concat+sort+compact fold and binary-search "
                                "tombstone filter (oracle/ii2_oracle.c); this run is also the correctness check of the GPU result.  "
                                "The pool's time is the serial critical path of the largest terms (every fold re-sorts the growing list, "
                                "file/types.go:14-22; the rank-1 term alone folds 16 lists into ~70M ids on one core), not %d cores of "
                                "work.  value_1_thread: terms [%d, %d) (%d postings in, %.1f s) on one thread" % (avail, t1_lo, T, n1, c1)}
    host_offs = offs
    del vals
    steps = args.merge_steps or min(args.steps, 10)
    dt, dev_ms, dev_n = time_merges(job, segs, tomb, out_off, out_vals, steps)
    alg = merge_alg_bytes(segs, T, k, args.docs, int(st.n_out))
    step_s = dev_ms / max(dev_n, 1) * 1e-3
    traffic, traffic_src = (None, None)
    if T == 1_000_000 and k == 16 and args.merge_mean == 1000.0 and args.docs == 100_000_000:
        traffic, traffic_src = pmc_traffic("merge")
    res = {
        "value": n_in * steps / dt, "unit": "postings/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "scaling": "n/a (N = 1)",
        "config": {"workload": "%d-way segment merge, %d terms x mean %.0f postings, 10 %% duplicated postings, 1 %% tombstones "
                               "(BASELINE configs[2])" % (k, T, args.merge_mean),
                   "postings_in": n_in, "postings_out": int(st.n_out), "terms_out": int(st.n_terms_out), "tiles": int(st.n_tiles),
                   "generate_s": round(gen_s, 1)},
        "roofline": {"bound": "hbm", "kernel": "one ii2_merge_segments call: every kernel from the first decode launch to the "
                                               "final offset scan (names in profiles/)",
                     "achieved": alg / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / step_s / 1e9 / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": int(alg),
                     "kernel_avg_us": step_s * 1e6, "launches_timed": int(dev_n),
                     "timing": "one HIP event pair per call around all of its kernels; wall ms_per_step adds the call's host side"},
    }
    if cpu is not None:
        res["cpu_baseline"] = cpu
    # Shard.Merge's REAL output (shard.go:207 -> file/writer.go:32-59): the merged terms as an encoded segment —
    # ii2_merge_segments_to_seg = the same merge + the one-pass DV1 encoder (encode_stream.hip).  Checked: the segment decodes to the
    # raw merge (which the oracle checked above); byte-for-byte equality with the oracle's encoding is tests/test_gpu_configs.py.
    try:
        mseg, st_s = ctx.merge_to_segment(segs, tomb)
        minfo = mseg.info
        po_s, v_s = mseg.decode()
        ok_s = int(st_s.n_out) == int(st.n_out) and bool(np.array_equal(po_s, out_off.download())) and \
            bool(np.array_equal(v_s, out_vals.download(int(st.n_out))))
        del po_s, v_s
        mseg.free()
        job.sync_all()
        t0s = time.perf_counter()
        for _ in range(steps):
            sgi, _ = ctx.merge_to_segment(segs, tomb)
            sgi.free()
        job.sync_all()
        seg_s = (time.perf_counter() - t0s) / steps
        enc_in = sum(sg.info.n_bytes for sg in segs) + 8 * sum(sg.info.n_blocks for sg in segs) + 4 * k * (T + 1)
        alg_s = enc_in + args.docs // 8 + int(minfo.n_bytes) + 8 * int(minfo.n_blocks) + 4 * (T + 1)     # SURVEY §8 d, enc_bytes(out) form
        traffic_s, traffic_s_src = pmc_traffic("merge_to_segment")
        res["to_segment"] = {
            "value": n_in / seg_s, "unit": "postings/s", "ms_per_step": seg_s * 1e3, "steps": steps,
            "ratio_to_raw_merge": seg_s / (dt / steps), "out_payload_bytes": int(minfo.n_bytes), "out_blocks": int(minfo.n_blocks),
            "roofline": {"bound": "hbm", "achieved": alg_s / seg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_s / seg_s / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": int(alg_s), "traffic": traffic_s, "traffic_source": traffic_s_src,
                         "timing": "wall clock per ii2_merge_segments_to_seg call (two host waits inside: the merge's counts, the encoder's byte count)"},
            "what": "the drop-in Shard.Merge path: merge + one-pass DV1 encode of the merged terms into a ready segment",
            "check": "decodes to the oracle-checked raw merge" if ok_s else "MISMATCH"}
        if not ok_s:
            job.rc = 5
    except Exception as e:  # noqa: BLE001
        res["to_segment"] = {"error": repr(e)}
    # end to end with the term alignment on the device: every segment keeps only the terms it really holds
    # (its own dictionary of 8-byte big-endian term ids); ii2_align_terms merges the k dictionaries, ii2_seg_select_aligned
    # builds the aligned views, then the same merge runs on them
    n_out_ref = int(st.n_out)
    chk_ref = int(out_vals.download(min(n_out_ref, 1 << 22)).astype(np.uint64).sum())
    try:
        dict_ids, compact = [], []
        for o, sg in zip(host_offs, segs):
            present = np.flatnonzero(np.diff(o.astype(np.int64)) > 0)
            dict_ids.append(present.astype(">u8"))
            po, vv = sg.decode()
            compact.append(ctx.encode(np.concatenate([[0], np.cumsum(np.diff(po.astype(np.int64))[present])]).astype(np.uint64), vv))
            del po, vv
        n_all = sum(d.size for d in dict_ids)
        # every segment's dictionary is made resident once, when the segment is (ii2_dict_create): the alignment reads them in HBM
        t0 = time.perf_counter()
        res_dicts = [ctx.dictionary_flat(np.frombuffer(d.tobytes(), np.uint8), np.arange(d.size + 1, dtype=np.uint64) * np.uint64(8)) for d in dict_ids]
        job.torch.cuda.synchronize()
        dict_ms = (time.perf_counter() - t0) * 1e3
        al = ctx.align_dicts(res_dicts)
        al.free()                                                 # warm-up (workspace growth)
        job.torch.cuda.synchronize()
        t0 = time.perf_counter()
        al = ctx.align_dicts(res_dicts)
        t1 = time.perf_counter()
        views = ctx.select_aligned_all(compact, al)
        t2 = time.perf_counter()
        _, _, st2 = ctx.merge(views, tomb, out_off, out_vals)
        t3 = time.perf_counter()
        ok = int(st2.n_out) == n_out_ref and al.n_union == T and \
            int(out_vals.download(min(n_out_ref, 1 << 22)).astype(np.uint64).sum()) == chk_ref
        res["end_to_end_with_device_alignment"] = {
            "dictionary_terms_in": int(n_all), "union_terms": int(al.n_union), "align_ms": (t1 - t0) * 1e3, "select_views_ms": (t2 - t1) * 1e3,
            "merge_ms": (t3 - t2) * 1e3, "total_ms": (t3 - t0) * 1e3, "value": n_in / (t3 - t0), "unit": "postings/s",
            "make_dictionaries_resident_ms": dict_ms,
            "note": "dictionaries resident in HBM (made once per segment, ii2_dict_create: make_dictionaries_resident_ms, not part of total_ms); "
                    "align_ms = ii2_align_dicts (ranking kernel + prefix sum + numbering); one run after one warm-up, wall clock",
            "check": "same result as the pre-aligned merge" if ok else "MISMATCH"}
        if not ok:
            job.rc = 5
        for v in views + compact + res_dicts:
            v.free()
        al.free()
    except Exception as e:  # noqa: BLE001 — the timed figures above stand on their own
        res["end_to_end_with_device_alignment"] = {"error": repr(e)}
    for s in segs:
        s.free()
    out_off.free()
    out_vals.free()
    return res


def bench_merge_strong(job):
    """configs[3]: ONE 64-segment x 1M-term merge, the terms cut into `world` contiguous ranges balanced by estimated
    merge cost; every rank merges its range chunk by chunk into DV1 segments (ii2_merge_segments_to_seg) and the merged
    segments are exchanged encoded (ii2_seg_allgather on a second context: chunk i travels while chunk i + 1 merges)."""
    args, ctx, world, rank = job.args, job.ctx, job.world, job.rank
    from concurrent.futures import ThreadPoolExecutor
    from inverted_index_2_amd import sharding, synth
    T, k = args.merge_terms, args.strong_segments
    ranges = sharding.balanced_term_ranges(T, args.merge_mean, args.docs, world, by="cost+encode")      # (merge -> DV1 segment)
    t0, t1 = ranges[rank]
    pretend = os.environ.get("BENCH_PRETEND")          # "r/N" on one GPU: rank r's share of an N-rank run, without the peers
    if pretend and world == 1:                          # (what one rank of the N-GPU job computes; scripts/README.md)
        pr, pn = (int(x) for x in pretend.split("/"))
        t0, t1 = sharding.balanced_term_ranges(T, args.merge_mean, args.docs, pn, by="cost+encode")[pr]
    offs, vals, removed = synth.merge_workload_big(T, k, args.merge_mean, args.docs, threads=min(cores_available(), 32),
                                                   term_range=(t0, t1))
    n_in_local = int(sum(int(o[-1]) for o in offs))
    segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
    tomb = ctx.tombstones(removed)
    out_off = ctx.empty(t1 - t0 + 1, np.uint64)
    out_vals = ctx.empty(max(n_in_local, 1))
    _, _, st = ctx.merge(segs, tomb, out_off, out_vals)
    # cheap exactness check of this rank's share (config 4 itself is oracle-checked in tests/test_gpu_configs.py):
    # per term ascending + duplicate-free + no tombstoned id, and the count of distinct (term, doc) pairs minus removed
    g_off = out_off.download().astype(np.int64)
    g_vals = out_vals.download(int(st.n_out))
    d = np.diff(g_vals.astype(np.int64))
    inner = np.ones(d.size, bool)
    b = g_off[1:-1]
    inner[b[(b > 0) & (b <= d.size)] - 1] = False
    if d.size and not np.all(d[inner] > 0):
        raise SystemExit(f"rank {rank}: merged lists are not strictly ascending")
    if np.isin(g_vals[:: max(1, g_vals.size // 4_000_000)], removed).any():
        raise SystemExit(f"rank {rank}: a tombstoned id survived the merge")
    del vals
    steps = args.merge_steps or min(args.steps, 10)
    dt_raw, dev_ms, dev_n = time_merges(job, segs, tomb, out_off, out_vals, steps)     # the raw-output merge of the whole range, for reference
    out_off.free()
    out_vals.free()

    # ---- the range in chunks (contiguous term sub-ranges of about equal estimated cost), each chunk's k views made once
    per_term = np.zeros(t1 - t0, np.int64)
    for o in offs:
        per_term += np.diff(o.astype(np.int64))
    del offs
    n_chunks = max(1, min(args.strong_chunks, t1 - t0))
    cum = np.concatenate([[0.0], np.cumsum(sharding.merge_cost_weights(per_term, encode=True))])
    shares = np.array([1.0] * (n_chunks - 1) + [min(max(args.strong_last_share, 0.05), 1.0)]) if n_chunks > 1 else np.array([1.0])
    bounds = np.cumsum(shares) / shares.sum()
    cuts = [0] + [int(np.searchsorted(cum, cum[-1] * bounds[c - 1])) for c in range(1, n_chunks)] + [t1 - t0]
    cuts = [max(a, b) for a, b in zip(cuts, np.maximum.accumulate(cuts))]
    views = []
    for a, b2 in zip(cuts[:-1], cuts[1:]):
        idx = np.arange(a, b2, dtype=np.int64)
        views.append([ctx.select(sg, idx) for sg in segs] if b2 > a else None)
    # the exchange runs on a second context of the same GPU (own stream, own communicator): the two only share segments
    from inverted_index_2_amd import Context
    xctx = Context(ctx.device)
    impl = "ii2_seg_allgather on one rank (a device copy)"
    if world > 1:
        job.init_comm(xctx)
        impl = ("ii2_seg_allgather: ncclAllGather of the shapes + ONE grouped ncclSend/ncclRecv exchange of the segment's six arrays "
                "per chunk over xGMI, two host waits per chunk" if job.comm == "ii2" else "torch.distributed.all_gather of the padded DV1 arrays (ii2_comm_init failed on some rank)")

    # where this rank's lists sit inside a gathered segment: after the lists of the ranks before it (every chunk holds one
    # list slot per term of the rank's chunk, so the counts are known without asking)
    chunk_lists = [[0] * world for _ in range(n_chunks)]
    if world > 1:
        mine_l = job.torch.tensor([b2 - a for a, b2 in zip(cuts[:-1], cuts[1:])], dtype=job.torch.int64, device=job.dev)
        all_l = [job.torch.zeros_like(mine_l) for _ in range(world)]
        job.dist.all_gather(all_l, mine_l)
        chunk_lists = [[int(all_l[r][c].item()) for r in range(world)] for c in range(n_chunks)]

    def exchange(seg, ci=0, check=False):
        """One chunk's merged segment to every rank; returns (bytes this rank received, own part intact).  check: the
        gathered segment's slice of THIS rank's lists is decoded and compared with the chunk's own decode - offsets, ids and
        therefore the block numbers and byte offsets the receiver shifted (outside the timed passes: it moves the ids to the host)."""
        if seg is None:
            seg_bytes = 0
        else:
            inf = seg.info
            seg_bytes = int(inf.n_bytes + 8 * inf.n_blocks + 4 * inf.n_lists)
        if world == 1 and seg is None:
            return 0, True
        if os.environ.get("BENCH_STUCK"):          # (rehearsal of the watchdog: an exchange that never comes back)
            time.sleep(3600)
        if world == 1 or job.comm == "ii2":
            if seg is None:          # a chunk in which nothing survived still takes part in the collective
                seg = empty_seg
            s0 = xctx.counters()[2]
            tq = time.perf_counter()
            g = xctx.seg_allgather(seg)
            if os.environ.get("BENCH_TRACE"):
                print("rank %d: seg_allgather %.2f ms, %d host waits" % (rank, (time.perf_counter() - tq) * 1e3, xctx.counters()[2] - s0), file=sys.stderr, flush=True)
            ginf = g.info
            got = int(ginf.n_bytes + 8 * ginf.n_blocks + 4 * ginf.n_lists)
            ok = ginf.n_postings >= seg.info.n_postings and ginf.n_lists >= seg.info.n_lists
            if check and ok and seg is not empty_seg:
                first = sum(chunk_lists[ci][:rank]) if world > 1 else 0
                n_l = int(seg.info.n_lists)
                gpo, gv = g.decode()
                mpo, mv = seg.decode()
                gpo = gpo.astype(np.int64)
                lo_p, hi_p = int(gpo[first]), int(gpo[first + n_l])
                ok = bool(np.array_equal(gpo[first:first + n_l + 1] - lo_p, mpo.astype(np.int64)) and np.array_equal(gv[lo_p:hi_p], mv))
                del gpo, gv, mpo, mv
            g.free()
            return got, bool(ok)
        # fallback (the library communicator failed somewhere): the three arrays through torch's communicator, padded
        torch, dist = job.torch, job.dist
        blk, skip, payload = seg.export() if seg is not None else (np.zeros(1, np.uint32), np.zeros(1, np.uint64), np.zeros(0, np.uint8))
        raw = np.concatenate([blk.view(np.uint8), skip.view(np.uint8), payload.view(np.uint8)])
        n = torch.tensor([raw.size], dtype=torch.int64, device=job.dev)
        ns = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(ns, n)
        cap = max(int(x.item()) for x in ns)
        mine = torch.from_numpy(np.concatenate([raw, np.zeros(cap - raw.size, np.uint8)])).to(job.dev)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        ok = bool(torch.equal(parts[rank][: raw.size].cpu(), torch.from_numpy(raw)))
        return int(sum(int(x.item()) for x in ns)), ok

    empty_seg = ctx.encode_lists([np.empty(0, np.uint32)]) if world > 1 else None

    # the chunks are merged by `strong-workers` threads side by side, one context each (the segments and the tombstones belong
    # to the device, not to a context): one chunk's host round trips (sizes of the new segment, its mirrors) hide behind
    # another chunk's kernels.  Chunks still reach the exchange in term order.
    n_workers = max(1, min(args.strong_workers, n_chunks))
    wctx = [ctx] + [Context(ctx.device) for _ in range(n_workers - 1)]
    wpool = ThreadPoolExecutor(max_workers=n_workers)

    def merge_chunk(ci, vw):
        if vw is None:
            return None
        tq = time.perf_counter()
        sg, _ = wctx[ci % n_workers].merge_to_segment(vw, tomb)
        if os.environ.get("BENCH_TRACE"):
            print("rank %d: merge_to_segment %.2f ms (worker %d)" % (rank, (time.perf_counter() - tq) * 1e3, ci % n_workers), file=sys.stderr, flush=True)
        return sg

    def one_pass(with_exchange, check=False):
        """All chunks of this rank: merged to segments by the workers, handed to the exchange thread in term order."""
        pending, merged = [], []
        futs = [wpool.submit(merge_chunk, ci, vw) for ci, vw in enumerate(views)]
        for ci, f in enumerate(futs):
            sg = f.result()
            merged.append(sg)
            if with_exchange:
                pending.append(pool.submit(exchange, sg, ci, check))
        tq = time.perf_counter()
        got = [f.result() for f in pending]
        tr = time.perf_counter()
        for sg in merged:
            if sg is not None:
                sg.free()
        if os.environ.get("BENCH_TRACE"):
            print("rank %d: waited %.2f ms for the exchanges, freed in %.2f ms" % (rank, (tr - tq) * 1e3, (time.perf_counter() - tr) * 1e3), file=sys.stderr, flush=True)
        return got

    pool = ThreadPoolExecutor(max_workers=1)
    # correctness of the chunked, encoded path, once: the chunks' merged segments decode to the raw merge of the range
    chk_ok = True
    pos = 0
    for vw, a, b2 in zip(views, cuts[:-1], cuts[1:]):
        if vw is None:
            continue
        sg, stc = ctx.merge_to_segment(vw, tomb)
        n_c = int(g_off[b2] - g_off[a])
        if sg is None:
            chk_ok &= n_c == 0
            continue
        po, vv = sg.decode()
        keep = np.diff(g_off[a:b2 + 1]) > 0                   # the encoder keeps every term slot; empty ones have no postings
        chk_ok &= vv.size == n_c and np.array_equal(vv, g_vals[int(g_off[a]):int(g_off[b2])]) and \
            np.array_equal(np.diff(po.astype(np.int64)) > 0, keep)
        sg.free()
        pos += n_c
    if not chk_ok:
        print(f"rank {rank}: the chunked merge-to-segment path differs from the raw merge", file=sys.stderr, flush=True)
        job.rc = 5
    # warm-up (staging buffers and, N > 1, the communicator's channels) - and the exchange's content check: every gathered
    # segment's slice of this rank's lists decodes to the chunk that was sent.  The first collective of the run: under the watchdog.
    warm = job.guarded("the merge's first segment exchange", lambda: one_pass(True, check=True))
    warm_ok = all(ok for _, ok in warm)
    if not warm_ok:
        print(f"rank {rank}: a gathered segment does not hold this rank's chunk", file=sys.stderr, flush=True)
        job.rc = 4
    job.sync_all()
    t_a = time.perf_counter()
    for _ in range(steps):
        one_pass(False)
    job.sync_all()
    dt_merge = job.max_over_ranks(time.perf_counter() - t_a)
    exch_ok, recv_bytes = True, 0

    def timed_with_exchange():
        nonlocal exch_ok, recv_bytes
        t_b = time.perf_counter()
        for _ in range(steps):
            for got, ok in one_pass(True):
                recv_bytes += got
                exch_ok &= ok
        job.sync_all()
        return time.perf_counter() - t_b
    dt_x = job.max_over_ranks(job.guarded("the merge's segment exchange", timed_with_exchange))
    pool.shutdown()
    wpool.shutdown()
    for c in wctx[1:]:
        c.close()
    tot = job.torch.tensor([float(n_in_local), float(st.n_out)], dtype=job.torch.float64, device=job.dev)
    if world > 1:
        job.dist.all_reduce(tot)
    n_in_total, n_out_total = int(tot[0].item()), int(tot[1].item())
    res = {
        "value": n_in_total * steps / dt_x, "unit": "postings/s", "ms_per_step": dt_x / steps * 1e3, "steps": steps, "scaling": "strong",
        "what": "whole job per step: every rank merges its term range chunk by chunk into DV1 segments and every chunk is exchanged "
                "encoded with all ranks (overlapped with the next chunk's merge) - the exchange is INSIDE the timed region",
        "config": {"workload": "ONE %d-way segment merge of %d terms x mean %.0f postings (BASELINE configs[3]), terms cut into "
                               "%d contiguous ranges balanced by estimated merge + encode cost (sharding.merge_cost_weights), %d chunks per rank merged by "
                               "%d worker contexts" % (k, T, args.merge_mean, world, n_chunks, n_workers),
                   "postings_in_total": n_in_total, "postings_out_total": n_out_total,
                   "rank0_terms": [int(t0), int(t1)] if rank == 0 else None, "rank0_postings_in": n_in_local if rank == 0 else None,
                   "parallelism": "terms%d" % world},
        "merge_to_segment_only": {"value": n_in_total * steps / dt_merge, "ms_per_step": dt_merge / steps * 1e3,
                                  "what": "the same chunked merge + encode without the exchange"},
        "raw_output_merge": {"value": n_in_total * steps / dt_raw, "ms_per_step": dt_raw / steps * 1e3, "rank0_device_ms_per_step": dev_ms / max(dev_n, 1),
                             "what": "one ii2_merge_segments call per rank over its whole range (u32 CSR out), no exchange"},
        "exchange": {"impl": impl, "bytes_received_per_rank_per_step": recv_bytes // max(steps, 1),
                     "bytes_if_raw_u32": 4 * n_out_total,
                     "check": ("chunks decode to the raw merge; this rank's slice of every gathered segment decodes to the chunk it sent "
                               "(offsets and ids, checked once before the timed passes)" if (exch_ok and chk_ok and warm_ok)
                               else "MISMATCH on rank %d" % rank)},
    }
    if not exch_ok:
        print(f"rank {rank}: segment exchange mismatch", file=sys.stderr, flush=True)
        job.rc = 4
    for vw in views:
        for v in (vw or []):
            v.free()
    if empty_seg is not None:
        empty_seg.free()
    for sg in segs:
        sg.free()
    xctx.close()
    return res


C5_RANKS = (2, 4, 16, 64, 256, 1024, 4096, 16384)
C5_GEN_SHARDS = 8


def c5_lists(D, world, rank):
    """This rank's doc range of the config-5 index.  The index is defined on 8 fixed doc shards (every (term, shard) list
    has its own seed), so the data is the same whether 1, 2, 4 or 8 ranks hold it; a rank generates only its shards."""
    from inverted_index_2_amd import sharding, synth
    n_gen = C5_GEN_SHARDS if C5_GEN_SHARDS % world == 0 else world
    lo, hi = sharding.doc_range(rank, world, D)
    rng = np.random.default_rng(55)
    core = np.unique(rng.integers(0, D, 10_000)).astype(np.uint32)
    core = sharding.slice_list_to_docs(core, lo, hi)
    lists = []
    for zr in C5_RANKS:
        parts = []
        for g in range(n_gen):
            a, b = sharding.doc_range(g, n_gen, D)
            if a >= lo and b <= hi:
                parts.append(synth.geometric_postings(1.0 / zr, b - a, synth.term_seed(zr * 64 + g), a))
        parts.append(core)
        lists.append(np.unique(np.concatenate(parts)).astype(np.uint32))
    return lists, (lo, hi)


def c5_roofline(info, n_out, step_s, D, world):
    """SURVEY §8 d: the intersection's algorithmic bytes are the WHOLE encoded lists + skip tables + the result ("a galloping
    kernel that skips blocks may touch fewer bytes; report touched bytes separately, keep this formula for the roofline").  For
    this query the formula says little about the kernel: it gallops - 10,000 result ids out of 833M postings - and touches a few
    percent of the bytes; `touched_bytes` (FETCH_SIZE + WRITE_SIZE of one query, profiles/r04_pmc_c5.json) is what moves."""
    alg = int(info.n_bytes + 8 * info.n_blocks + 4 * n_out)
    touched, src = (None, None)
    if D == 1_000_000_000 and world == 1:
        touched, src = pmc_traffic("c5")
    return {"bound": "latency (dependent block fetches; see DESIGN.md §4.1b) - the HBM formula is kept as SURVEY §8 d prescribes",
            "achieved": alg / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / step_s / 1e9 / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": alg, "touched_bytes": touched, "touched_bytes_source": src,
            "touched_frac_of_algorithmic": (touched / alg) if touched else None,
            "touched_GBps": (touched / step_s / 1e9) if touched else None}


def bench_c5(job):
    """configs[4]: one 8-term conjunctive query over a 1B-doc index, doc-range sharded, all-gatherv in rank order."""
    args, ctx, world, rank = job.args, job.ctx, job.world, job.rank
    from oracle import oracle as orc
    D = args.c5_docs
    if D > (1 << 32):
        raise SystemExit("c5: the doc-id space is uint32")
    g0 = time.perf_counter()
    lists, (lo, hi) = c5_lists(D, world, rank)
    gen_s = time.perf_counter() - g0
    n_in = int(sum(l.size for l in lists))
    seg = ctx.encode_lists(lists)
    sel = [(seg, i) for i in range(len(lists))]
    cap = int(min(l.size for l in lists)) + 512
    out = ctx.empty(cap)
    d_count = ctx.empty(8, np.uint64)
    _, n_out = ctx.intersect(sel, out=out)
    got = out.download(n_out)
    t_o = time.perf_counter()
    want = orc.intersect(lists[::-1])                     # the oracle on this rank's doc range (shortest list first)
    oracle_s = time.perf_counter() - t_o
    if n_out != want.size or not np.array_equal(got, want):
        raise SystemExit(f"rank {rank}: the 8-term intersection differs from the oracle on doc range [{lo}, {hi})")
    steps = args.c5_steps or min(args.steps, 20)
    for _ in range(min(args.warmup, 5)):
        ctx.intersect_async(sel, None, out, d_count)
    job.sync_all()
    ctx.profile_region(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.intersect_async(sel, None, out, d_count)
    ctx.profile_region(False)
    job.sync_all()
    dt = job.max_over_ranks(time.perf_counter() - t0)
    dev_s = ctx.profile_region_ms() * 1e-3
    tot = job.torch.tensor([float(n_in), float(n_out), float(int(got.astype(np.uint64).sum()) % (1 << 52))], dtype=job.torch.float64, device=job.dev)
    if world > 1:
        job.dist.all_reduce(tot)
    n_in_total, n_out_total, sum_total = int(tot[0].item()), int(tot[1].item()), int(tot[2].item())
    info = seg.info
    res = {
        "value": n_in_total * steps / dt, "unit": "postings/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
        "scaling": "strong (one query, the doc-id space cut into %d ranges)" % world,
        "config": {"workload": "8-term AND, Zipf ranks %s over a %d-doc index with a 10,000-id common core (BASELINE configs[4]), "
                               "doc-range sharded (sharding.doc_range), DV1 decoded in-kernel, galloping over the skip tables"
                               % (list(C5_RANKS), D),
                   "postings_in_total": n_in_total, "result_ids_total": n_out_total, "rank0_doc_range": [int(lo), int(hi)] if rank == 0 else None,
                   "rank0_list_lengths": [int(l.size) for l in lists] if rank == 0 else None, "parallelism": "docrange%d" % world,
                   "generate_s": round(gen_s, 1)},
        "rank0_device_us_per_query": dev_s / steps * 1e6,
        "rank0_encoded_bytes": int(info.n_bytes + 8 * info.n_blocks),
        "roofline": c5_roofline(info, n_out, dev_s / steps, D, world),
        "check": "every rank's result equals the oracle's on its doc range (rank 0: %d ids, oracle %.2f s)" % (n_out, oracle_s),
    }
    if world > 1:
        # the exchange inside the timed region: query + rank-order concatenation of the results, per step
        job.init_comm()
        capx = job.torch.tensor([cap], dtype=job.torch.int64, device=job.dev)
        job.dist.all_reduce(capx, op=job.dist.ReduceOp.MAX)

        def with_exchange():
            good, total = True, 0
            job.sync_all()
            t1 = time.perf_counter()
            for it in range(steps):
                _, n1 = ctx.intersect(sel, out=out)
                gout, counts, impl = job.gather(out, n1, int(capx.item()))
                total = int(sum(counts))
                if it == steps - 1:
                    allv = gout.download(total)
                    good = bool(np.all(np.diff(allv.astype(np.int64)) > 0) and total == n_out_total and
                                int(allv.astype(np.uint64).sum()) % (1 << 52) == sum_total % (1 << 52) and
                                np.array_equal(allv[sum(counts[:rank]):sum(counts[:rank + 1])], got))
                gout.free()
            job.sync_all()
            return time.perf_counter() - t1, total, impl, good
        dtx, total, impl, good = job.guarded("c5's all-gatherv", with_exchange)
        dtx = job.max_over_ranks(dtx)
        res["with_allgatherv"] = {"value": n_in_total * steps / dtx, "ms_per_step": dtx / steps * 1e3, "ids": total, "impl": impl,
                                  "check": ("ascending, the ranks' counts and id checksum add up, this rank's ids at its rank offset"
                                            if good else "MISMATCH on rank %d" % rank)}
        if not good:
            print(f"rank {rank}: c5 all-gatherv mismatch", file=sys.stderr, flush=True)
            job.rc = 4
    seg.free()
    out.free()
    d_count.free()
    return res


def assemble(job, parts):
    """The one JSON line from the bench objects measured so far (the headline is the first of intersect / merge / merge_strong / c5)."""
    args, world = job.args, job.world
    head = parts.get("intersect") or parts.get("merge") or parts.get("merge_strong") or parts["c5"]
    result = {
        "metric": METRIC, "value": head["value"], "unit": "postings/s", "n_gpus": world, "steps": args.steps if "intersect" in parts else head.get("steps", args.steps),
        "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
        "scaling": "weak" if "intersect" in parts else head.get("scaling", "weak"), "vs_baseline": None, "dtype": "u32",
        "data": "synthetic", "config": head["config"],
    }
    for key, val in head.items():          # everything else the headline workload reports (roofline, cpu_baseline, ...)
        if key not in result:
            result[key] = val
    if "intersect" in parts and world > 1:
        result["scaling_note"] = ("the headline is WEAK scaling (every rank intersects its own %d-doc universe, nothing is exchanged in the "
                                  "timed region); the strong-scaling figures are merge_strong.value and c5.value" % args.docs)
    for name in ("merge", "merge_strong", "c5"):
        if name in parts and parts[name] is not head:
            result[name] = parts[name]
    return result


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if args.dry_run:
        sys.exit(dry_run(args))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != max(args.gpus, 1):
        print("bench.py: --gpus %d but the launcher started %s ranks; the line reports the ranks that run" % (args.gpus, os.environ["WORLD_SIZE"]),
              file=sys.stderr, flush=True)
    job = Job(args)
    world, rank = job.world, job.rank
    parts = job.parts
    if args.workload in ("all", "intersect"):
        parts["intersect"] = bench_intersect(job)

    def secondary(name, fn):
        """A bench object besides the headline: what goes wrong in it is reported in its place (and in the exit code), the
        line with everything else is still printed."""
        try:
            parts[name] = fn(job)
        except SystemExit:
            raise
        except Exception as e:  # noqa: BLE001
            print("rank %d: %s failed: %r" % (rank, name, e), file=sys.stderr, flush=True)
            if not parts:
                raise
            if world > 1:
                # the other ranks are inside this object's collectives: this rank cannot go on to the next object without
                # crossing them.  Rank 0 leaves what it has measured on stdout; the launcher ends the job.
                if rank == 0:
                    line = assemble(job, dict(parts))
                    line["aborted"] = "%s failed on rank 0: %r; objects measured after it are missing" % (name, e)
                    print(json.dumps(line), flush=True)
                raise
            parts[name] = {"error": repr(e)}
            job.rc = max(job.rc, 6)
    if args.workload in ("all", "merge") and world == 1:
        secondary("merge", bench_merge)
    if args.workload in ("all", "strong") or (args.workload == "merge" and world > 1):
        secondary("merge_strong", bench_merge_strong)
    if args.workload in ("all", "c5"):
        secondary("c5", bench_c5)
    result = assemble(job, parts)
    if job.rehearse:
        result["rehearsal"] = "all %d ranks shared cuda:0 (gloo control plane, torch exchange fallback): control-flow check only" % world
    if world > 1:
        rc = job.torch.tensor([job.rc], dtype=job.torch.int32, device=job.dev)
        job.dist.all_reduce(rc, op=job.dist.ReduceOp.MAX)
        job.rc = int(rc.item())
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        try:
            job.dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
    sys.exit(job.rc)


if __name__ == "__main__":
    main()
