#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of `bench.py` into profiles/pmc_summary.json.

    python tools/pmc_summary.py <dir with *_counter_collection.csv files> <examples_per_gpu> [workload]

Per kernel (matched by name) the counters are averaged over its dispatches.  HBM bytes follow the
gfx950 recipe: FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE tallies the 128-byte requests of
wide (16 B per lane) reads at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte stores.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d, B = sys.argv[1], int(sys.argv[2])
    workload = sys.argv[3] if len(sys.argv) > 3 else "miniboone"
    acc = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> per-dispatch values
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                key = (path, row["Dispatch_Id"], row["Counter_Name"])
                per_dispatch[key] += float(row["Counter_Value"])
                names[(path, row["Dispatch_Id"])] = row["Kernel_Name"]
        for (pth, disp, ctr), v in per_dispatch.items():
            acc[names[(pth, disp)]][ctr].append(v)
    out = {"workload": workload, "examples_per_gpu": B, "source": "rocprofv3 --pmc passes of bench.py (see profiles/*_pmc_*.csv)", "kernels": {}}
    for k, ctrs in acc.items():
        short = k.split("(")[0][:80]
        ent = {c: sum(v) / len(v) for c, v in ctrs.items()}
        ent["dispatches"] = max(len(v) for v in ctrs.values())
        if "FETCH_SIZE" in ent and "WRITE_SIZE" in ent:
            ent["hbm_read_bytes"] = 2.0 * ent["FETCH_SIZE"] * 1024
            ent["hbm_write_bytes"] = ent["WRITE_SIZE"] * 1024
            ent["hbm_bytes"] = ent["hbm_read_bytes"] + ent["hbm_write_bytes"]
        out["kernels"][short] = ent
    # the encoder launch is the fused-MLP instance with 64 outputs: 2 output tiles of 32 in the split-bf16
    # kernel (the default arithmetic), 4 tiles of 16 in the exact-f32 kernel
    # (the 16-row split kernel, the default: 4 output tiles of 16, 8-wave workgroups)
    # the default arithmetic: the fp16 split (two terms per operand); a run that also times the other arithmetics
    # holds their instances too
    for k, ent in out["kernels"].items():
        if "cs_mlp_split16_kernel<16, 4, 8, 2>" in k:
            out["encoder_launch"] = ent
            out["mlp_arith"] = "f16x2"
    if "encoder_launch" not in out:
        for k, ent in out["kernels"].items():
            if "cs_mlp_split16_kernel<16, 4, 8, 3>" in k:
                out["encoder_launch"] = ent
                out["mlp_arith"] = "bf16x3"
    if "encoder_launch" not in out:
        for k, ent in out["kernels"].items():
            if "cs_mlp_split_kernel<8, 2>" in k:
                out["encoder_launch"] = ent
                out["mlp_arith"] = "bf16x3"
    if "encoder_launch" not in out:
        for k, ent in out["kernels"].items():
            if "cs_mlp_kernel<16, 4, 4>" in k:
                out["encoder_launch"] = ent
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_summary.json"), "w"),
              indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
