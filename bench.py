#!/usr/bin/env python3
"""bench.py — posting-list hot path on MI355X: postings/s and fraction of the HBM roofline.

Workload at N=1 (BASELINE.json configs[1]): 2-term intersection over a 100M-doc index, Zipf
ranks 2 and 3 (≈50M and ≈33M postings), block-Δ-varint (DV1) lists resident in HBM, decoded
in-kernel; output = the ascending doc ids of the intersection, resident in HBM.
A step = one pass of the hot path (partition pre-pass + tile kernel) over that input.

N > 1 (one process per GPU, launched by torch.distributed.run): the doc-id space is sharded —
rank g holds the lists' postings in [g*D, (g+1)*D) — so per-GPU work is fixed (weak scaling)
and ranks do not talk during the timed steps (the reference's shards are independent,
inverted_index.go:83-103).  The rank-order concatenation of the per-rank results
(RCCL all-gatherv, inverted_index.go:330-339) runs once after the timed region and is
reported separately as `allgatherv_ms` (use --gather-timed to put it inside every step).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--docs", type=int, default=100_000_000, help="doc-id universe per GPU (config 2: 100M)")
    ap.add_argument("--workload", choices=["intersect", "merge"], default="intersect")
    ap.add_argument("--tombstones", action="store_true", help="apply a 1%% tombstone bitmap during the intersection")
    ap.add_argument("--gather-timed", action="store_true", help="include the RCCL all-gatherv in every timed step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather-timeout", type=int, default=120, help="seconds the post-timing all-gatherv may take (N > 1)")
    ap.add_argument("--merge-terms", type=int, default=200_000, help="merge workload: aligned terms")
    ap.add_argument("--merge-segments", type=int, default=16)
    ap.add_argument("--merge-mean", type=float, default=1000.0)
    return ap.parse_args()


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


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from inverted_index_2_amd import Context, comm_unique_id, synth

    ctx = Context(local_rank)
    ctx.selftest()

    if args.workload == "merge":
        return bench_merge(args, ctx, torch, dist, world, rank)

    D = args.docs
    offset = rank * D
    if offset + D > (1 << 32):
        raise SystemExit("doc-id shards exceed the uint32 id space")
    a = synth.zipf_list(2, D, offset)
    b = synth.zipf_list(3, D, offset)
    removed = None
    tomb = None
    if args.tombstones:
        removed = (synth.geometric_postings(0.01, D, synth.term_seed(10**6), offset)).astype(np.uint32)
        tomb = ctx.tombstones(removed)
    seg = ctx.encode_lists([a, b])
    lists = [(seg, 0), (seg, 1)]
    n_in = int(a.size + b.size)
    out = ctx.empty(min(a.size, b.size) + 512)
    d_count = ctx.empty(8, np.uint64)

    # correctness of this rank's result before timing: count + order + checksum
    _, n_out = ctx.intersect(lists, tomb=tomb, out=out)
    got = out.download(n_out)
    want_np = np.intersect1d(a, b, assume_unique=True)
    if removed is not None:
        want_np = np.setdiff1d(want_np, removed, assume_unique=True)
    if n_out != want_np.size or not np.array_equal(got, want_np):
        raise SystemExit(f"rank {rank}: GPU intersection differs from the numpy cross-check")

    gather_out = None
    gstate = {"impl": None}

    def init_comm():
        # the library's own RCCL communicator (ii2_comm_*); torch's communicator only carries the unique id
        nonlocal gather_out
        if gstate["impl"] is not None:
            return
        gather_out = ctx.empty((min(a.size, b.size) + 512) * world)
        try:
            uid = [comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            ctx.comm_init(world, rank, uid[0])
            gstate["impl"] = "ii2_allgatherv (ncclAllGather of counts + grouped ncclSend/ncclRecv)"
        except Exception as e:  # noqa: BLE001 — keep the scaling run alive; the exchange then goes through torch (RCCL too)
            gstate["impl"] = "torch.distributed.all_gather fallback (%s)" % type(e).__name__

    if world > 1 and args.gather_timed:
        init_comm()

    def gather_all():
        if gstate["impl"].startswith("ii2_"):
            return ctx.allgatherv(out, n_out, gather_out, world)
        # fallback: padded all_gather through torch's RCCL communicator, then pack on the host side of the check
        cap = min(a.size, b.size) + 512
        cnt = torch.tensor([n_out], dtype=torch.int64, device="cuda")
        cnts = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(cnts, cnt)
        mine = torch.from_numpy(np.concatenate([got, np.zeros(cap - n_out, np.uint32)]).view(np.int32)).cuda()
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        counts = [int(c.item()) for c in cnts]
        packed = np.concatenate([pp[:c].cpu().numpy().view(np.uint32) for pp, c in zip(parts, counts)])
        gather_out.upload(packed)
        return counts

    def step():
        ctx.intersect_async(lists, tomb, out, d_count)
        if args.gather_timed and world > 1:
            gather_all()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.set_option("profile.events", 8)      # every 8th pass is bracketed by HIP events (a pair idles the stream ~10 us)
    ctx.profile_read()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    kern_ms, kern_n = ctx.profile_read()
    ctx.set_option("profile.events", 0)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # secondary figure: two passes in flight (two contexts = two HIP streams).  The partition / expand
    # kernels of one pass then overlap the tile kernel of the other; same work per step.
    pipelined = None
    if world == 1:
        ctx2 = Context(local_rank)
        seg2 = ctx2.encode_lists([a, b])
        out2 = ctx2.empty(min(a.size, b.size) + 512)
        cnt2 = ctx2.empty(8, np.uint64)
        tomb2 = ctx2.tombstones(removed) if removed is not None else None
        pair = [(ctx, lists, tomb, out, d_count), (ctx2, [(seg2, 0), (seg2, 1)], tomb2, out2, cnt2)]
        for c, ls, tb, o, dc in pair:
            c.intersect_async(ls, tb, o, dc)
        torch.cuda.synchronize()
        p0 = time.perf_counter()
        for i in range(args.steps):
            c, ls, tb, o, dc = pair[i & 1]
            c.intersect_async(ls, tb, o, dc)
        torch.cuda.synchronize()
        pdt = time.perf_counter() - p0
        if not np.array_equal(out2.download(n_out), got):
            raise SystemExit("second stream's result differs")
        pipelined = {"streams": 2, "value": n_in * args.steps / pdt, "unit": "postings/s", "ms_per_step": pdt / args.steps * 1e3}
        ctx2.close()

    info = seg.info
    # algorithmic bytes of one pass (SURVEY.md §8 d): encoded payload + 8 B per block of
    # skip table + 4 B per result id (+ D/8 tombstone bitmap).  The HIP events bracket the whole pass
    # on the library's stream: k_isect_partition + k_isect_tiles (dominant) + k_isect_expand.
    alg_bytes = info.n_bytes + 8 * info.n_blocks + 4 * n_out + (D // 8 if tomb is not None else 0)
    kern_avg_s = (kern_ms / max(kern_n, 1)) * 1e-3
    achieved = alg_bytes / kern_avg_s / 1e9 if kern_n else None
    # HBM traffic from the PMC passes kept under profiles/ (same command, same workload); null otherwise
    traffic = None
    try:
        if D == 100_000_000 and tomb is None and world == 1:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_intersect.json")) as f:
                traffic = json.load(f)["hbm_bytes_per_pass_corrected"]
    except OSError:
        pass

    result = {
        "metric": "postings/sec (intersect + segment-merge) at 1/2/4/8 MI355X; % HBM roofline",
        "value": n_in * world * args.steps / dt,
        "unit": "postings/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {
            "workload": "2-term intersection (Zipf ranks 2 and 3), %d-doc universe per GPU, DV1 block-delta-varint "
                        "decoded in-kernel, doc-range sharded" % D,
            "postings_per_gpu": n_in, "result_ids_per_gpu": n_out, "tombstones": bool(args.tombstones),
            "encoded_bytes_per_gpu": int(info.n_bytes), "blocks_per_gpu": int(info.n_blocks),
            "parallelism": "docrange%d" % world, "allgatherv": "timed" if args.gather_timed else "after timed region",
        },
        "roofline": {
            "bound": "hbm", "kernel": "one pass: ii2::k_isect_partition + ii2::k_isect_tiles (dominant) + ii2::k_isect_expand",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
            "algorithmic_bytes_per_launch": int(alg_bytes), "kernel_avg_us": kern_avg_s * 1e6, "launches_timed": int(kern_n),
        },
    }
    if pipelined is not None:
        result["pipelined"] = pipelined
    # the exchange step, once, outside the timed region: rank-order concatenation of the results.  The timed
    # figures above are already final; a watchdog keeps a stuck exchange from taking the whole scaling run with it.
    if world > 1:
        import threading

        def on_timeout():
            if rank == 0:
                result["allgatherv_impl"] = "skipped: the exchange did not finish within %d s" % args.gather_timeout
                print(json.dumps(result), flush=True)
            os._exit(0)

        timer = threading.Timer(args.gather_timeout, on_timeout)
        timer.daemon = True
        timer.start()
        try:      # nothing in here may cost the run its (already final) timed figures
            init_comm()
            torch.cuda.synchronize()
            g0 = time.perf_counter()
            counts = gather_all()
            gather_ms = (time.perf_counter() - g0) * 1e3
            total_out = sum(counts)
            allv = gather_out.download(total_out)
            good = bool(np.all(np.diff(allv.astype(np.int64)) > 0) and np.array_equal(allv[sum(counts[:rank]):sum(counts[:rank + 1])], got))
            result["allgatherv_ms"] = gather_ms
            result["allgatherv_impl"] = gstate["impl"]
            result["allgatherv_ids"] = int(total_out)
            result["allgatherv_check"] = "rank-order concatenation verified on rank 0" if good else "MISMATCH on rank %d" % rank
            if not good:
                print(f"rank {rank}: all-gatherv result is not the rank-order concatenation", file=sys.stderr, flush=True)
        except BaseException as e:  # noqa: BLE001
            result["allgatherv_impl"] = "failed: %s: %s" % (type(e).__name__, e)
        timer.cancel()
        if rank != 0:
            try:
                dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass
            return
    if not args.no_cpu_baseline and world == 1:      # the CPU baseline is a rank-0, N=1 figure
        reps = 20                                      # ~8 s single-thread + ~8 s on the pool
        v1, res, per1 = cpu_baseline_intersect([a, b], removed, reps, threads=1)
        if not np.array_equal(res, got):
            raise SystemExit("GPU result differs from the oracle")
        ncores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
        vN, resN, perN = cpu_baseline_intersect([a, b], removed, max(reps * max(1, ncores // 2), reps), threads=ncores)
        if not np.array_equal(resN, got):
            raise SystemExit("GPU result differs from the oracle (sharded run)")
        result["cpu_baseline"] = {
            "value": vN, "unit": "postings/s", "cores": ncores, "kind": "port", "value_1_thread": v1,
            "sample": "the full rank-0 workload (DV1 decode + two-pointer intersection of %d postings) cut into %d doc-range "
                      "shards on a %d-thread pool, %.3f s per repetition; single thread: %.2f s per repetition, %d repetitions "
                      "(oracle/ii2_oracle.c)" % (n_in, ncores, ncores, perN, per1, reps),
        }
    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def bench_merge(args, ctx, torch, dist, world, rank):
    """Secondary workload (BASELINE config 3 family): k-way segment merge with tombstones.
    Terms are sharded over the ranks in contiguous ranges; each rank generates its own shard."""
    from inverted_index_2_amd import synth
    T, k = args.merge_terms, args.merge_segments
    offs, vals, removed = synth.merge_workload(T, k, args.merge_mean, args.docs, seed=synth.GLOBAL_SEED + rank)
    segs = [ctx.encode(o, v) for o, v in zip(offs, vals)]
    tomb = ctx.tombstones(removed)
    n_in = int(sum(int(o[-1]) for o in offs))
    out_off = ctx.empty(T + 1, np.uint64)
    out_vals = ctx.empty(n_in)
    _, _, st = ctx.merge(segs, tomb, out_off, out_vals)
    for _ in range(args.warmup):
        ctx.merge(segs, tomb, out_off, out_vals)
    ctx.set_option("profile.events", 1)
    ctx.profile_read()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.merge(segs, tomb, out_off, out_vals)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    kern_ms, kern_n = ctx.profile_read()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    enc = sum(s.info.n_bytes for s in segs)
    nblk = sum(s.info.n_blocks for s in segs)
    alg = enc + 8 * nblk + 4 * k * (T + 1) + args.docs // 8 + 4 * st.n_out + 4 * (T + 1)
    kavg = kern_ms / max(kern_n, 1) * 1e-3
    # HBM traffic of one merge (all its kernels) from the PMC passes kept under profiles/ (same command, default sizes)
    traffic = None
    try:
        if T == 200_000 and k == 16 and args.merge_mean == 1000.0 and args.docs == 100_000_000 and world == 1:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_merge.json")) as f:
                traffic = json.load(f)["hbm_bytes_per_pass_corrected"]
    except OSError:
        pass
    result = {
        "metric": "postings/sec (intersect + segment-merge) at 1/2/4/8 MI355X; % HBM roofline",
        "value": n_in * world * args.steps / dt, "unit": "postings/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "%d-way segment merge, %d terms x mean %.0f postings per GPU, 1%% tombstones, term-sharded"
                               % (k, T, args.merge_mean), "postings_in_per_gpu": n_in, "postings_out_per_gpu": int(st.n_out),
                   "tiles": int(st.n_tiles), "parallelism": "terms%d" % world},
        "roofline": {"bound": "hbm", "kernel": "ii2::k_merge_tiles", "achieved": alg / kavg / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": alg / kavg / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": int(alg), "kernel_avg_us": kavg * 1e6, "launches_timed": int(kern_n)},
    }
    # BASELINE config 4's exchange step, once, after the timed region: the ranks' merged postings concatenated in rank
    # (= term-range) order with the library's RCCL all-gatherv.  A watchdog keeps a stuck exchange from taking the run.
    gather = None
    if world > 1:
        import threading
        from inverted_index_2_amd import comm_unique_id
        done = threading.Event()

        def on_timeout():
            if not done.is_set():
                if rank == 0:
                    result["allgatherv_impl"] = "skipped: the exchange did not finish within %d s" % args.gather_timeout
                    print(json.dumps(result), flush=True)
                os._exit(0)

        timer = threading.Timer(args.gather_timeout, on_timeout)
        timer.daemon = True
        timer.start()
        try:
            counts_t = torch.tensor([int(st.n_out)], dtype=torch.int64, device="cuda")
            all_counts = [torch.zeros_like(counts_t) for _ in range(world)]
            dist.all_gather(all_counts, counts_t)
            total = int(sum(int(c.item()) for c in all_counts))
            gout = ctx.empty(total + 8)
            uid = [comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            ctx.comm_init(world, rank, uid[0])
            torch.cuda.synchronize()
            g0 = time.perf_counter()
            counts = ctx.allgatherv(out_vals, int(st.n_out), gout, world)
            gather = {"allgatherv_ms": (time.perf_counter() - g0) * 1e3, "allgatherv_ids": int(sum(counts)),
                      "allgatherv_impl": "ii2_allgatherv (ncclAllGather of counts + grouped ncclSend/ncclRecv)"}
            mine = gout.download(int(sum(counts[:rank + 1])))[int(sum(counts[:rank])):]
            good = bool(np.array_equal(mine, out_vals.download(int(st.n_out))))
            gather["allgatherv_check"] = "rank-order concatenation verified on rank 0" if good else "MISMATCH on rank %d" % rank
        except BaseException as e:  # noqa: BLE001 — the timed figures stand on their own
            gather = {"allgatherv_impl": "failed: %s: %s" % (type(e).__name__, e)}
        done.set()
        timer.cancel()
    if rank != 0:
        return
    if gather:
        result.update(gather)
    if not args.no_cpu_baseline and world == 1:
        from oracle import oracle as orc
        ncores = min(len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1), 16)
        t0 = time.perf_counter()
        w_off, w_vals, _ = orc.merge_segments(offs, vals, removed, threads=ncores)
        cdt = time.perf_counter() - t0
        if not (np.array_equal(out_off.download(), w_off) and np.array_equal(out_vals.download(int(w_off[-1])), w_vals)):
            raise SystemExit("GPU merge differs from the oracle")
        result["cpu_baseline"] = {"value": n_in / cdt, "unit": "postings/s", "cores": ncores, "kind": "port",
                                  "sample": "the full rank-0 merge workload once, oracle worker pool over term ranges"}
    print(json.dumps(result))


if __name__ == "__main__":
    main()
