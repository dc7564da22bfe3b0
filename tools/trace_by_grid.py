"""Per-(kernel, grid) durations from a rocprofv3 kernel trace: `python tools/trace_by_grid.py <..._kernel_trace.csv> [skip_first_n_dispatches]`.
Separates launches of one template at different shapes (the decoder's w_1 / w_2 / linear_q all run gemm_f16x2_kernel<2, 2, ...>), which the
--stats table averages together. Prints launches, mean / min / max microseconds and total milliseconds per group, largest total first."""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[skip:]
    groups = collections.defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"]
        name = name[:70]
        grid = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])),
                int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_Z"])))
        groups[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
    out = sorted(groups.items(), key=lambda kv: -sum(kv[1]))
    print(f"{'kernel':70s} {'workgroups':>18s} {'n':>6s} {'mean us':>9s} {'min':>8s} {'max':>8s} {'total ms':>9s}")
    for (name, grid), d in out[:60]:
        print(f"{name:70s} {str(grid):>18s} {len(d):6d} {sum(d) / len(d):9.1f} {min(d):8.1f} {max(d):8.1f} {sum(d) / 1000:9.2f}")


if __name__ == "__main__":
    main()
