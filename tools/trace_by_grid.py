"""Average duration per (kernel, grid size) from a rocprofv3 --kernel-trace CSV, for the kernels matching a substring:
    python tools/trace_by_grid.py <kernel_trace.csv> <substring> [out.csv]
(bench.py launches the same GEMM instantiation at two shapes: 16 images per launch in the synchronous / profiled
regions, several batches per launch in the pipelined timed region; --stats averages them together.)"""
import collections
import csv
import sys


def main():
    path, sub = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if sub not in r["Kernel_Name"]:
                continue
            grid = int(r.get("Grid_Size", 0) or 0) or int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
            key = (r["Kernel_Name"].split("(")[0][:90], grid)
            acc[key][0] += 1
            acc[key][1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    rows = [("Name", "Grid_Size", "Calls", "AverageNs", "TotalDurationNs")]
    for (name, grid), (n, tot) in sorted(acc.items()):
        rows.append((name, grid, n, "%.1f" % (tot / n), "%.0f" % tot))
    out = open(sys.argv[3], "w", newline="") if len(sys.argv) > 3 else sys.stdout
    csv.writer(out).writerows(rows)


if __name__ == "__main__":
    main()
