#!/usr/bin/env python3
"""Copy the results of tools/profile_bench.sh (gpurun_out/prof_r3*) into profiles/ under a tag and rebuild
profiles/pmc_summary.json.

    python tools/collect_profiles.py r3b
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
pairs = [("prof_r3/stats/r3_kernel_stats.csv", f"{tag}_lair_miniboone_kernel_stats.csv"),
         ("prof_r3_bench.json", f"{tag}_lair_miniboone_bench_under_rocprof.json"),
         ("prof_r3/fetch/r3_counter_collection.csv", f"{tag}_pmc_fetch_size.csv"),
         ("prof_r3/write/r3_counter_collection.csv", f"{tag}_pmc_write_size.csv"),
         ("prof_r3/sq/r3_counter_collection.csv", f"{tag}_pmc_sq_grbm.csv")]
for a, b in pairs:
    shutil.copyfile(os.path.join(src, a), os.path.join(dst, b))
    print(b)
out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), os.path.join(src, "prof_r3"), "65536"],
                     capture_output=True, text=True)
print(out.stdout[-2000:], out.stderr[-2000:])
