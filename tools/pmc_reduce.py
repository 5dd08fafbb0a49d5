#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc passes into profiles/pmc_traffic.json entries.

    python tools/pmc_reduce.py KEY KERNEL_SUBSTR FETCH_DIR WRITE_DIR [OUT_JSON]

Each DIR is the -d directory of one `rocprofv3 --pmc <COUNTER> --kernel-trace` pass (FETCH_SIZE, WRITE_SIZE:
one counter per pass, as MI355X_MICROARCH.md prescribes). Values are averaged over the dispatches whose kernel
name contains KERNEL_SUBSTR. Corrections (same guide): both counters are in KiB; on gfx950 FETCH_SIZE counts
half the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is taken as is.
"""
import csv, glob, json, os, sys


def mean_counter(d, counter, kernel):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter and kernel in row.get("Kernel_Name", ""):
                    vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for kernel ~{kernel!r} under {d}")
    return sum(vals) / len(vals), len(vals)


def main():
    key, kernel, fdir, wdir = sys.argv[1:5]
    out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(os.path.dirname(__file__), "..", "profiles", "pmc_traffic.json")
    fetch_kb, nf = mean_counter(fdir, "FETCH_SIZE", kernel)
    write_kb, nw = mean_counter(wdir, "WRITE_SIZE", kernel)
    rd, wr = fetch_kb * 1024 * 2, write_kb * 1024
    doc = json.load(open(out)) if os.path.exists(out) else {}
    doc[key] = {"fetch_size_kb_raw": fetch_kb, "write_size_kb_raw": write_kb, "hbm_read_bytes": round(rd), "hbm_write_bytes": round(wr),
                "hbm_bytes_per_launch": round(rd + wr), "dispatches": [nf, nw]}
    json.dump(doc, open(out, "w"), indent=1)
    print(key, doc[key])


if __name__ == "__main__":
    main()
