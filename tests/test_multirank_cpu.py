"""CPU, world_size 2 over gloo: the host logic of the N > 1 path — inverted_index_2_amd/sharding.py (doc-range and
balanced term-range partitions) and the product library's exchange arithmetic ii2_gatherv_offsets (csrc/comm.cpp) —
in a real two-process run.  The per-rank posting work is done with plain numpy set operations here (no GPU, no
oracle): what is under test is that the shares partition the problem and that the rank-order concatenation at the
computed offsets is the global answer (inverted_index.go:330-339).  The RCCL transport itself (ncclSend/ncclRecv in
ii2_allgatherv) needs GPUs and is NOT executed by this test."""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from inverted_index_2_amd import _lib, sharding, synth


def _gatherv(local: np.ndarray, world: int, rank: int) -> np.ndarray:
    """all-gatherv over gloo laid out by the library's offset function (the same arithmetic ii2_allgatherv uses)."""
    cnt = torch.tensor([local.size], dtype=torch.int64)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    counts = (C.c_uint64 * world)(*[int(c.item()) for c in cnts])
    off = (C.c_uint64 * (world + 1))()
    lib = _lib.load()
    assert lib.ii2_gatherv_offsets(counts, world, sum(counts), off) == 0
    assert lib.ii2_gatherv_offsets(counts, world, sum(counts) - 1, off) == -4 or sum(counts) == 0      # ECAPACITY on every rank alike
    cap = max(int(c) for c in counts)
    mine = torch.zeros(max(cap, 1), dtype=torch.int64)
    mine[: local.size] = torch.from_numpy(local.astype(np.int64))
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    out = np.zeros(int(off[world]), np.uint32)
    for r in range(world):
        out[int(off[r]):int(off[r + 1])] = parts[r][: int(counts[r])].numpy().astype(np.uint32)
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # --- one conjunctive query, doc-range sharded (BASELINE config 5 layout) ---
        D = 400_000
        lists = [synth.zipf_list(r, D) for r in (2, 3, 7)]
        lo, hi = sharding.doc_range(rank, world, D)
        sl = [sharding.slice_list_to_docs(l, lo, hi) for l in lists]
        local = sl[0]
        for x in sl[1:]:
            local = np.intersect1d(local, x, assume_unique=True)
        got = _gatherv(local.astype(np.uint32), world, rank)
        want = lists[0]
        for x in lists[1:]:
            want = np.intersect1d(want, x, assume_unique=True)
        assert np.array_equal(got, want), "doc-range sharded AND"
        assert np.all(np.diff(got.astype(np.int64)) > 0)
        # --- segment merge, term ranges balanced by posting count (BASELINE config 4 layout) ---
        T, k, D2 = 4096, 4, 50_000
        ranges = sharding.balanced_term_ranges(T, 40.0, D2, world)
        assert ranges[0][0] == 0 and ranges[-1][1] == T and all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:]))
        t0, t1 = ranges[rank]
        offs, vals, removed = synth.merge_workload_big(T, k, 40.0, D2, threads=2, term_range=(t0, t1))

        def merged(offs, vals):
            out = []
            for t in range(offs[0].size - 1):
                u = np.unique(np.concatenate([v[int(o[t]):int(o[t + 1])] for o, v in zip(offs, vals)]))
                out.append(np.setdiff1d(u, removed, assume_unique=True))
            return out
        mine = merged(offs, vals)
        got = _gatherv(np.concatenate(mine).astype(np.uint32) if mine else np.empty(0, np.uint32), world, rank)
        f_offs, f_vals, f_removed = synth.merge_workload_big(T, k, 40.0, D2, threads=2)
        assert np.array_equal(f_removed, removed)
        for s in range(k):      # a rank's share is exactly its slice of the full workload
            assert np.array_equal(f_vals[s][int(f_offs[s][t0]):int(f_offs[s][t1])], vals[s])
        want = np.concatenate(merged(f_offs, f_vals)).astype(np.uint32)
        assert np.array_equal(got, want), "term-range sharded merge"
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc() + repr(e)))
    finally:
        dist.destroy_process_group()


def test_world2_sharding_and_concatenation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_key_ranges_partition_the_keyspace():
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            lo, hi = sharding.key_range(r, world)
            seen.extend(range(lo, hi))
        assert seen == list(range(sharding.N_SHARD_KEYS))
        assert sharding.owner_of_key(1023, world) == world - 1
    # shardKey (shard.go:362-378) known answers: len < 2 -> 0; top 10 bits of the first two bytes
    for t, want in ((b"", 0), (b"a", 0), (b"aa", (0x6161 >> 6)), (b"term1", (0x7465 >> 6)), (b"\xff\xff", 1023), (b"zz", 0x7A7A >> 6)):
        assert sharding.shard_key(t) == want


def test_balanced_term_ranges_balance_postings():
    sizes, fine = synth.merge_chunk_bounds(1_000_000, 1000.0, 100_000_000)
    for world in (2, 4, 8):
        rs = sharding.balanced_term_ranges(1_000_000, 1000.0, 100_000_000, world, by="postings")
        assert rs[0][0] == 0 and rs[-1][1] == 1_000_000 and all(a[1] == b[0] for a, b in zip(rs[:-1], rs[1:]))
        assert all(a in fine and b in fine for a, b in rs)
        share = [sizes[a:b].sum() / sizes.sum() for a, b in rs]
        assert max(share) < 1.1 / world and min(share) > 0.9 / world
        # the default balances estimated merge cost: small terms weigh more per posting, so the tail ranks hold fewer postings
        rc = sharding.balanced_term_ranges(1_000_000, 1000.0, 100_000_000, world)
        assert rc[0][0] == 0 and rc[-1][1] == 1_000_000 and all(a[1] == b[0] for a, b in zip(rc[:-1], rc[1:]))
        assert all(a in fine and b in fine for a, b in rc)
        w = sharding.merge_cost_weights(sizes)
        cshare = [w[a:b].sum() / w.sum() for a, b in rc]
        assert max(cshare) < 1.1 / world and min(cshare) > 0.9 / world
        assert sizes[rc[-1][0]:].sum() < sizes[rs[-1][0]:].sum()
