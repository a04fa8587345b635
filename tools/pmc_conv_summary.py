#!/usr/bin/env python3
"""Derived per-kernel figures from the output of tools/pmc_conv.sh (averaged counters per kernel).

    python tools/pmc_conv_summary.py gpurun_out/conv_all/omniglot_pmc.txt > profiles/<tag>_conv_omniglot_pmc_summary.txt

matrix-pipe occupancy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); other vector instructions
per matrix instruction = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA; LDS bank-conflict share =
SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; HBM bytes = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB (gfx950: FETCH_SIZE tallies the
128-byte requests of wide reads at 64 bytes).
"""
import re
import sys

txt = open(sys.argv[1]).read()
print("# " + __doc__.strip().replace("\n", "\n# "))
for blk in txt.split("== ")[1:]:
    name = blk.split("\n")[0]
    vals = {m.group(1): float(m.group(2)) for m in re.finditer(r"(\w+)\s+([\d.]+)\s+\(", blk)}
    if "GRBM_GUI_ACTIVE" not in vals:
        continue
    mf = vals.get("SQ_INSTS_MFMA", 0.0)
    occ = vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024 / (vals["GRBM_GUI_ACTIVE"] / 8)
    other = (vals.get("SQ_INSTS_VALU", 0.0) - mf) / mf if mf else float("nan")
    bc = vals.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(vals.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0)
    hbm = (2 * vals.get("FETCH_SIZE", 0.0) + vals.get("WRITE_SIZE", 0.0)) * 1024 / 1e6
    print(f"== {name}: matrix-pipe occupancy {occ * 100:.1f}%, other vector / matrix instructions {other:.2f}, "
          f"LDS bank-conflict share {bc * 100:.1f}%, HBM {hbm:.0f} MB per launch, {vals['GRBM_GUI_ACTIVE'] / 8 / 1e3:.0f}k shader cycles")
    for k in sorted(vals):
        print(f"   {k:32s} {vals[k]:16.1f}")
