#!/usr/bin/env python3
"""Writes tests/golden/ref_kats.json — the known-answer tables held by the reference's own
tests, transcribed by hand as DATA (inputs + expected outputs).  Nothing here is computed:
every expected value below is a literal from the cited reference test.  Go map iteration
order is random in IngestBulkCmd, so puts are listed in ascending value order (the
expected results do not depend on it).

Script ops (target "shard" = one Shard, "index" = InvertedIndex):
  ["put", [terms...], val]
  ["read", min|null, max|null, [[term, [vals...]], ...]]        expected full result, in order
  ["merge", reqCount, mCount, expected_merged|null]              shard.Merge / index.Merge(…, concurrency=2)
  ["remove", [vals...]]                                          Shard.Remove / InvertedIndex.PutRemoved
  ["segments", n]                                                number of live segments
  ["removed_values", [vals...]]                                  RemovedLists.Values()
  ["prefix", [prefixes...], {prefix: [vals...]}]
  ["shards", n]
"""
import json
import os

KATS = {
    # shard_test.go:40-63
    "TestInitFromExistingFiles": {"target": "shard", "script": [
        ["put", ["term1", "term2"], 1],
        ["put", ["term2", "term3"], 2],
        ["read", None, None, [["term1", [1]], ["term2", [1, 2]], ["term3", [2]]]],
    ]},
    # shard_test.go:65-88
    "TestIngestion": {"target": "shard", "script": [
        ["put", ["term1"], 1],
        ["read", None, None, [["term1", [1]]]],
        ["put", ["term1"], 1],
        ["put", ["term1", "term2"], 2],
        ["put", ["term3"], 3],
        ["read", None, None, [["term1", [1, 2]], ["term2", [2]], ["term3", [3]]]],
    ]},
    # shard_test.go:90-136 (merged variant, Merge(2,200))
    "TestReadPartial_merged": {"target": "shard", "script": [
        ["put", ["AA"], 1], ["put", ["BB"], 2], ["put", ["CC"], 3],
        ["merge", 2, 200, None],
        ["read", "AA", "BB", [["AA", [1]], ["BB", [2]]]],
        ["read", "BB", "CC", [["BB", [2]], ["CC", [3]]]],
    ]},
    # shard_test.go:90-136 (direct-segment variant)
    "TestReadPartial_direct": {"target": "shard", "script": [
        ["put", ["AA"], 1], ["put", ["BB"], 2], ["put", ["CC"], 3],
        ["read", "AA", "BB", [["AA", [1]], ["BB", [2]]]],
        ["read", "BB", "CC", [["BB", [2]], ["CC", [3]]]],
    ]},
    # shard_test.go:138-162
    "TestMerging": {"target": "shard", "script": [
        ["put", ["term1"], 1], ["put", ["term1", "term2"], 2], ["put", ["term3"], 3],
        ["segments", 3],
        ["merge", 3, 2, 2], ["segments", 2],
        ["merge", 2, 2, 2], ["segments", 1],
        ["merge", 2, 2, 0], ["segments", 1],
        ["read", None, None, [["term1", [1, 2]], ["term2", [2]], ["term3", [3]]]],
    ]},
    # shard_test.go:164-190
    "TestMergeWithRemoval": {"target": "shard", "script": [
        ["put", ["term1", "term3"], 1], ["put", ["term2"], 2], ["put", ["term3"], 3],
        ["segments", 3],
        ["merge", 2, 2, 2], ["segments", 2],
        ["remove", [2]],
        ["merge", 2, 2, 2], ["segments", 1],
        ["read", None, None, [["term1", [1]], ["term3", [1, 3]]]],
        ["remove", [10]],
        ["removed_values", [10]],
    ]},
    # shard_test.go:192-214
    "TestMergeEmptySegment": {"target": "shard", "script": [
        ["put", ["term1"], 1],
        ["put", ["term1"], 1],
        ["remove", [1]],
        ["merge", 2, 2, 2],
        ["segments", 0],
        ["read", None, None, []],
        ["remove", [2]],
    ]},
    # shard_test.go:216-248 (one pass of the script the 100 goroutines run)
    "TestConcurrentAccess_script": {"target": "shard", "script": [
        ["put", ["term1"], 1], ["put", ["term1", "term2"], 2], ["put", ["term3"], 3],
        ["merge", 2, 2, 2],
        ["read", None, None, [["term1", [1, 2]], ["term2", [2]], ["term3", [3]]]],
    ]},
    # inverted_index_test.go:59-82
    "TestPutRemove": {"target": "index", "script": [
        ["put", ["aaaa", "bbbb"], 1],
        ["put", ["aaaa", "bbbb"], 1],
        ["put", ["aaaa"], 2],
        ["remove", [1]],
        ["merge", 2, 3, None],
        ["read", None, None, [["aaaa", [2]]]],
    ]},
    # inverted_index_test.go:140-194
    "TestPut": {"target": "index", "script": [
        ["put", ["ab1", "ab2"], 1],
        ["put", ["ab2", "cd1"], 2],
        ["read", None, None, [["ab1", [1]], ["ab2", [1, 2]], ["cd1", [2]]]],
        ["shards", 2],
    ]},
    # inverted_index_test.go:196-221
    "TestSearchByPrefix": {"target": "index", "script": [
        ["put", ["a12"], 1], ["put", ["a13"], 1], ["put", ["a13"], 2], ["put", ["a20"], 3], ["put", ["a30"], 4],
        ["put", ["termA"], 5], ["put", ["termB"], 6], ["put", ["termC"], 7],
        ["prefix", ["a1"], {"a1": [1, 2]}],
        ["prefix", ["term", "unknown"], {"term": [5, 6, 7]}],
    ]},
    # inverted_index_test.go:222-281
    "TestReadScoped": {"target": "index", "script": [
        ["put", ["aa"], 1], ["put", ["bb"], 2], ["put", ["cc"], 3], ["put", ["dd"], 4],
        ["read", None, None, [["aa", [1]], ["bb", [2]], ["cc", [3]], ["dd", [4]]]],
        ["read", "a~", None, [["bb", [2]], ["cc", [3]], ["dd", [4]]]],
        ["read", None, "cc", [["aa", [1]], ["bb", [2]], ["cc", [3]]]],
        ["read", "bb", "cc", [["bb", [2]], ["cc", [3]]]],
    ]},
}

# removed_list_test.go:9-24 — batches put at t1 < t2, Sync([t2, t3]) with t3 > t2
REMOVED = {
    "TestRemovedLists": {
        "batches": [[1, [1, 5, 10]], [2, [2, 20, 30]]],
        "values": [1, 2, 5, 10, 20, 30],
        "sync_timestamps": [2, 3],
        "values_after_sync": [2, 20, 30],
    }
}

# file/writer_test.go:11-46 and :48-84 — lists that must survive encode -> decode verbatim
CODEC = {
    "TestWriter": [["term1", [10, 500, 300]], ["term2", []], ["term3", [66, 5513]]],
    "TestWriterDirect": [["term1", [10]], ["term2", [11]]],
}

# shard.go:362-378 worked by hand: ((t0 << 8) + t1) >> 6; len < 2 -> 0
SHARD_KEY = [["", 0], ["a", 0], ["aa", 389], ["ab1", 389], ["cd1", 397], ["term1", 465],
             ["AA", 261], ["zz", 489]]

if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    sk = SHARD_KEY
    with open(os.path.join(here, "ref_kats.json"), "w") as f:
        json.dump({"scripts": KATS, "removed": REMOVED, "codec": CODEC, "shard_key": sk}, f, indent=1, sort_keys=True)
    print("wrote ref_kats.json")
