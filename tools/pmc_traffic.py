#!/usr/bin/env python3
"""Turn a `rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum` pass of bench.py into profiles/pmc_traffic.json:
HBM-side bytes per launch of every kernel of the query path (last dispatch of each kernel = a timed step),
RDREQ x 128 B + WRREQ x 64 B (MI355X_MICROARCH.md, HBM section: FETCH_SIZE = RDREQ x 64 B reads half the bytes of a
streaming read on gfx950, WRITE_SIZE is exact), stamped with the hash of the library sources so that bench.py only
quotes it for the code it was measured on.

usage: pmc_traffic.py <rocprof output dir> <reads_per_step> <leaves> [threshold] [out.json]"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHORT = [("k_classify", "k_classify"), ("k_tail_records", "k_tail_records"), ("k_tile_bin", "k_tile_bin"),
         ("k_tile_test", "k_tile_test"), ("k_tile_plan", "k_tile_plan"), ("k_tile_assign", "k_tile_assign"),
         ("k_bucket_scatter", "k_bucket_scatter"), ("k_bucket_scan", "k_bucket_scan"), ("k_verify_rec", "k_verify_rec"),
         ("k_verify", "k_verify("), ("k_finalize", "k_finalize"), ("k_collect_open", "k_collect_open")]


def main() -> None:
    from phagefilter_amd._ffi import source_stamp
    d, reads, leaves = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    thr = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
    out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(ROOT, "profiles", "pmc_traffic.json")
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {d}")
    rows = collections.defaultdict(dict)
    for f in files:
        for r in csv.DictReader(open(f)):
            rows[(int(r["Dispatch_Id"]), r["Kernel_Name"])][r["Counter_Name"]] = float(r["Counter_Value"])
    # the dispatches of the LAST bucketed step: from the last k_classify<true, ...> (the deferring build) launch group up to
    # the next k_classify of any kind (bench.py ends with a small parity query on the direct kernel)
    disp = sorted(rows.items())
    is_cls = ["k_classify" in k for (_, k), _ in disp]
    last_start = max(i for i, ((_, k), _) in enumerate(disp) if "k_classify<true" in k and (i == 0 or not is_cls[i - 1]))
    end = last_start + 1
    while end < len(disp) and not (is_cls[end] and not is_cls[end - 1]):
        end += 1
    per = collections.defaultdict(int)
    for (_, k), v in disp[last_start:end]:
        name = next((s for s, pat in SHORT if pat in k), None)
        if name is None:
            continue
        per[name] += int(v.get("TCC_EA0_RDREQ_sum", 0) * 128 + v.get("TCC_EA0_WRREQ_sum", 0) * 64)
    js = {"reads_per_step": reads, "leaves": leaves, "threshold": thr, "source_stamp": source_stamp(),
          "hbm_bytes_per_launch": dict(per),
          "source": "rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum (own pass, --kernel-trace only) of bench.py; "
                    "bytes = RDREQ*128 B + WRREQ*64 B summed over the kernel's dispatches of the last step; "
                    "Infinity-Cache hits are included in the EA counters"}
    json.dump(js, open(out, "w"), indent=1)
    print(json.dumps(js))


if __name__ == "__main__":
    main()
