#!/usr/bin/env python3
"""Reduce the FETCH_SIZE / WRITE_SIZE passes that scripts/profile_decode.sh leaves under gpurun_out/prof_decode to one
record, gpurun_out/decode_traffic.json (copied to profiles/decode_traffic.json, which bench.py reads for
`roofline.traffic`).  Per launch of the headline shape: HBM bytes = 2 x FETCH_SIZE (on gfx950 the counter reports half
the bytes of a wide, 16 B per lane, coalesced read stream) + WRITE_SIZE, both counters in KiB, summed over the decode
kernel and its split merge."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "prof_decode")


def per_launch(kind):
    files = glob.glob(os.path.join(OUT, kind, "*", "*counter_collection.csv"))
    if not files:
        raise SystemExit(f"no counter file for {kind}")
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        name = r["Kernel_Name"]
        if "paged_decode" in name:
            acc["merge" if "merge" in name else "main"].append(float(r["Counter_Value"]))
    # the first dispatch of each is the warm-up on cold code: leave it out when there are more
    return {k: sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0] for k, v in acc.items()}


def main():
    commit = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("MSGL_COMMIT", "unknown")
    fetch, write = per_launch("fetch"), per_launch("write")
    fetch_kib = sum(fetch.values())
    write_kib = sum(write.values())
    rec = {
        "shape": {"layout": "interleaved", "batch": 256, "ctx": 8192, "page_size": 16, "hq": 32, "hkv": 8},
        "fetch_size_kib_per_launch": round(fetch_kib, 1), "write_size_kib_per_launch": round(write_kib, 1),
        "per_kernel_kib": {"fetch": {k: round(v, 1) for k, v in fetch.items()}, "write": {k: round(v, 1) for k, v in write.items()}},
        "traffic_bytes_per_launch": int(2 * fetch_kib * 1024 + write_kib * 1024),
        "algorithmic_bytes_per_launch": 2 * 8 * 128 * 2 * 256 * 8192,
        "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate runs) on tools/decode_bench, decode kernel + split merge; "
               "2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes (gfx950 FETCH_SIZE counts half of a 16 B/lane read stream)",
        "commit": commit,
    }
    path = os.path.join(ROOT, "gpurun_out", "decode_traffic.json")
    with open(path, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
