"""GPU idle time between kernels from a rocprofv3 --kernel-trace CSV: the union of the kernel intervals against the wall span of the
last `steps` repetitions of the step (split at the largest gaps is not needed: the whole trace's tail is analysed).
    python tools/gpu_idle_gaps.py <kernel_trace.csv> [tail_fraction=0.5]"""
import csv
import sys


def main():
    path = sys.argv[1]
    tail = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    cut = t1 - int((t1 - t0) * tail)
    rows = [r for r in rows if r[0] >= cut]
    span = max(r[1] for r in rows) - rows[0][0]
    busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    gaps = []
    for s, e, n in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    gaps.sort(reverse=True)
    print("kernels %d, span %.2f ms, busy (union) %.2f ms, idle %.2f ms = %.2f %%" % (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6,
                                                                                 100.0 * (span - busy) / span))
    print("largest gaps (us, kernel that follows):")
    for g, n in gaps[:12]:
        print("  %8.1f  %s" % (g / 1e3, n[:90]))
    small = sum(g for g, _ in gaps if g < 20000)
    print("gaps < 20 us: %d, total %.2f ms" % (sum(1 for g, _ in gaps if g < 20000), small / 1e6))


if __name__ == "__main__":
    main()
