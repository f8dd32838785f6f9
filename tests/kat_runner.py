"""Runs a golden script (tests/golden/ref_kats.json) against any object exposing the
reference's Shard / InvertedIndex operations — the oracle's ref_model or the product's
host mirror — the Python analogue of helper_test.go's TestingMachine."""


def _b(x):
    return None if x is None else x.encode()


def run_script(target, script, n_segments=None, removed_values=None, n_shards=None):
    for op in script:
        name = op[0]
        if name == "put":
            target.put([t.encode() for t in op[1]], op[2])
        elif name == "read":
            got = [(bytes(t), [int(v) for v in vals]) for t, vals in target.read(_b(op[1]), _b(op[2]))]
            want = [(t.encode(), vals) for t, vals in op[3]]
            assert got == want, (op, got)
        elif name == "merge":
            merged = target.merge(op[1], op[2]) if n_shards is None else target.merge(op[1], op[2], 2)
            if op[3] is not None:
                assert merged == op[3], (op, merged)
        elif name == "remove":
            (target.remove if n_shards is None else target.put_removed)(op[1])
        elif name == "segments":
            assert n_segments(target) == op[1], op
        elif name == "removed_values":
            assert [int(v) for v in removed_values(target)] == op[1], op
        elif name == "prefix":
            got = {bytes(k): [int(v) for v in vs] for k, vs in target.prefix_search([p.encode() for p in op[1]]).items()}
            assert got == {k.encode(): v for k, v in op[2].items()}, (op, got)
        elif name == "shards":
            assert n_shards(target) == op[1], op
        else:
            raise ValueError(name)
