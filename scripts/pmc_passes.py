"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output) into HBM bytes per PASS of a workload.

usage: pmc_passes.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <first-kernel-of-a-pass> <passes>

A pass = every dispatch from one dispatch of <first-kernel-of-a-pass> (name prefix) up to the next one; the LAST <passes>
passes of the run are averaged (warm-up and verification passes come first).  Every kernel dispatched inside a pass
counts — the library's own kernels, the hipcub scans and the runtime's fill / copy kernels alike.  Units and corrections
follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are KiB per dispatch; on gfx950 FETCH_SIZE
reports half of a wide coalesced read and is doubled; WRITE_SIZE is taken as is."""
import csv, json, sys
from collections import defaultdict


def short(name):
    n = name.replace("void ", "")
    for cut in ("<", "("):
        if cut in n:
            n = n.split(cut)[0]
    return n


def passes(path, counter, first, n_passes):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), float(r["Counter_Value"]) * 1024.0))
    rows.sort()
    starts = [i for i, (_, k, _) in enumerate(rows) if k.startswith(first)]
    assert len(starts) >= n_passes, (len(starts), n_passes)
    starts = starts[-n_passes:] + [len(rows)]
    per_kernel = defaultdict(float)
    for a, b in zip(starts[:-1], starts[1:]):
        for _, k, v in rows[a:b]:
            per_kernel[k] += v / n_passes
    return dict(per_kernel)


def main():
    fetch_csv, write_csv, out, first, n_passes = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
    extra = json.loads(sys.argv[6]) if len(sys.argv) > 6 else {}
    fe = passes(fetch_csv, "FETCH_SIZE", first, n_passes)
    wr = passes(write_csv, "WRITE_SIZE", first, n_passes)
    kernels, total = {}, 0.0
    for k in sorted(set(fe) | set(wr)):
        f_raw, w_raw = fe.get(k, 0.0), wr.get(k, 0.0)
        kernels[k] = {"FETCH_SIZE_raw_bytes_per_pass": f_raw, "FETCH_SIZE_corrected_bytes_per_pass": 2.0 * f_raw, "WRITE_SIZE_bytes_per_pass": w_raw}
        total += 2.0 * f_raw + w_raw
    res = {"units": "bytes per pass (counter KiB x 1024, summed over every dispatch of the pass, averaged over the last %d passes); "
                    "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read; random 4-byte reads are "
                    "uncalibrated and may be over-counted by this); WRITE_SIZE as is; Infinity-Cache hits are counted" % n_passes,
           "pass_starts_with": first, "kernels": kernels, "hbm_bytes_per_pass_corrected": total}
    res.update(extra)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({"hbm_bytes_per_pass_corrected": total, "top": sorted(((v["FETCH_SIZE_corrected_bytes_per_pass"] + v["WRITE_SIZE_bytes_per_pass"], k) for k, v in kernels.items()), reverse=True)[:8]}))


if __name__ == "__main__":
    main()
