"""Per-kernel totals of ONE step out of a rocprofv3 --kernel-trace csv: the kernels between the last two launches of a marker
kernel (one that runs once per step), grouped by name: launches, total time, idle gap before them.

usage: python tools/step_kernels.py <trace dir> <marker substring> [occurrences per step, default 1] [seq]
("seq": the launch sequence itself - start, duration, gap before - after the totals)
"""
import csv
import glob
import sys


def main():
    d, marker = sys.argv[1], sys.argv[2]
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(marks) < 2 * per:
        sys.exit(f"marker {marker!r}: {len(marks)} launches, need {2 * per}")
    a, b = marks[-2 * per], marks[-per]
    seq = rows[a:b]
    agg, prev = {}, rows[a - 1][1] if a else seq[0][0]
    for s, e, name in seq:
        k = name[:150] if "at::native" in name else name.split("(")[0][:70]
        n, t, g = agg.get(k, (0, 0, 0))
        agg[k] = (n + 1, t + (e - s), g + max(0, s - prev))
        prev = max(prev, e)
    busy = sum(e - s for s, e, _ in seq)
    print(f"# kernels {len(seq)}  span {(rows[b][0] - seq[0][0]) / 1e3:.1f} us  busy {busy / 1e3:.1f} us")
    for k, (n, t, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:4d} x  dur {t / 1e3:8.1f} us  avg {t / n / 1e3:6.1f}  gap-before {g / 1e3:7.1f}  {k}")
    if len(sys.argv) > 4 and sys.argv[4] == "seq":
        prev = rows[a - 1][1] if a else seq[0][0]
        for s, e, name in seq:
            print("%9.1f us  dur %7.1f  gap %6.1f  %s" % ((s - seq[0][0]) / 1e3, (e - s) / 1e3, max(0, s - prev) / 1e3, name.split("(")[0][:90]))
            prev = max(prev, e)


if __name__ == "__main__":
    main()
