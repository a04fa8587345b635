"""Per-launch times of one forward pass from a rocprofv3 kernel trace of tools/bench_conv.py <net> <rows> <enc|dec>.

    python tools/conv_trace_layers.py gpurun_out/prof_conv/conv_kernel_trace.csv

The trace of a run that repeats one forward pass is periodic: the last period is printed, one line per launch
(kernel, grid, duration), with the average over the last 10 periods.
"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    # the library's kernels only (the framework's own copies at start-up and tear-down are not periodic)
    rows = [r for r in rows if "cs_" in r["Kernel_Name"] or "fillBuffer" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    n = len(names)
    period = next(p for p in range(4, n // 3) if names[n - p:] == names[n - 2 * p:n - p] == names[n - 3 * p:n - 2 * p])
    reps = min(10, n // period - 1)
    total = 0.0
    for i in range(period):
        t = sum(dur[n - (k + 1) * period + i] for k in range(reps)) / reps / 1e3
        total += t
        r = rows[n - period + i]
        short = names[n - period + i].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        gx = int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))
        gy = int(r.get('Grid_Size_Y', 1)) // max(1, int(r.get('Workgroup_Size_Y', 1)))
        gz = int(r.get('Grid_Size_Z', 1)) // max(1, int(r.get('Workgroup_Size_Z', 1)))
        print(f"{i:3d} {short:46s} grid {gx:6d} x {gy} x {gz}  block {r['Workgroup_Size_X']:>4s}  {t:8.1f} us")
    print(f"period {period} launches, {total / 1e3:.3f} ms of kernel time per forward (average of the last {reps})")


if __name__ == "__main__":
    main()
