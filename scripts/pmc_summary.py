"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM bytes per dispatch.

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <kernel-prefix> [...]
Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): the counters are
KiB per dispatch; on gfx950 FETCH_SIZE reports half of a wide coalesced read and is doubled; WRITE_SIZE is taken
as is.  Only dispatches of kernels whose name starts with one of the prefixes are kept; the first `skip`
dispatches of each (warm-up, verification pass) are dropped via --skip N.
"""
import csv, json, sys
from collections import defaultdict

def per_kernel(path, counter, prefixes, skip):
    vals = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if any(name.startswith(p) for p in prefixes):
                vals[name].append(float(r["Counter_Value"]) * 1024.0)
    return {k: sum(v[skip:]) / max(len(v[skip:]), 1) for k, v in vals.items() if len(v) > skip}

def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    skip = 0
    for a in sys.argv[1:]:
        if a.startswith("--skip="):
            skip = int(a.split("=")[1])
    fetch_csv, write_csv, out = args[:3]
    prefixes = args[3:]
    fe = per_kernel(fetch_csv, "FETCH_SIZE", prefixes, skip)
    wr = per_kernel(write_csv, "WRITE_SIZE", prefixes, skip)
    kernels = {}
    total = 0.0
    for k in sorted(set(fe) | set(wr)):
        f_raw, w_raw = fe.get(k, 0.0), wr.get(k, 0.0)
        kernels[k] = {"FETCH_SIZE_raw_bytes": f_raw, "FETCH_SIZE_corrected_bytes": 2.0 * f_raw,
                      "WRITE_SIZE_raw_bytes": w_raw, "WRITE_SIZE_corrected_bytes": w_raw}
        total += 2.0 * f_raw + w_raw
    json.dump({"units": "bytes per dispatch (counter KiB x 1024); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports "
                        "half of a wide coalesced read); WRITE_SIZE as is", "skip_first_dispatches": skip,
               "kernels": kernels, "hbm_bytes_per_pass_corrected": total}, open(out, "w"), indent=1)
    print(json.dumps({"hbm_bytes_per_pass_corrected": total, "kernels": list(kernels)}))

if __name__ == "__main__":
    main()
