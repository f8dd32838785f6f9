"""GPU probe: segment file write / load rates of the host mirror's file layer (host/segment_file.h): one segment of T
terms x ~mean postings, written with file.Writer (device encode + export + two files) and read back with file.Reader
(files + device import + decode).  Usage: persist_probe.py [terms] [mean] [dir]"""
import sys, os, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from inverted_index_2_amd import Context, synth
from inverted_index_2_amd.host import SegmentFiles
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
mean = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
base = sys.argv[3] if len(sys.argv) > 3 else None
ctx = Context(0)
offs, vals, _ = synth.merge_workload_big(T, 1, mean, 100_000_000, dup_frac=0.0, threads=16)
post_off, values = offs[0], vals[0]
terms = np.arange(T, dtype=">u8").view(np.uint8)          # 8-byte big-endian term ids: byte order = numeric order
term_off = np.arange(T + 1, dtype=np.uint64) * 8
d = tempfile.mkdtemp(dir=base)
f = SegmentFiles(ctx)
try:
    for rep in range(2):
        t = time.perf_counter(); key = f.write_arrays(d, terms, term_off, post_off, values); tw = time.perf_counter() - t
        sz = sum(os.path.getsize(os.path.join(d, x)) for x in os.listdir(d) if x.startswith(key))
        t = time.perf_counter(); nt, _ = f.read_count(d, key); tr = time.perf_counter() - t
        assert nt == T
        print(f"terms {T} postings {values.size} files {sz/1e6:.1f} MB: write {tw*1e3:.0f} ms ({values.size/tw/1e6:.0f} M postings/s), "
              f"read+decode {tr*1e3:.0f} ms ({values.size/tr/1e6:.0f} M postings/s)", flush=True)
        f.remove(d, key)
finally:
    shutil.rmtree(d, ignore_errors=True)
f.close(); ctx.close()
