"""Timeline view of ONE train step from a rocprofv3 --kernel-trace CSV (side stream on: kernels overlap, so per-kernel sums do not
add up to wall time): step wall, union-busy time, idle gaps, per-kernel totals, and the launch sequence with gaps.
Usage: python tools/step_timeline.py <kernel_trace.csv> [marker=adam_kernel] [--seq]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+)<(.*)>\(.*", name)
    return (f"{m.group(1)}<{m.group(2)}>" if m else name.split("(")[0])[:90]


def main(path, marker="adam_kernel", seq=False):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Stream_Id", r.get("Queue_Id", "?"))))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[2].startswith(marker)]
    if len(marks) < 3:
        print("not enough steps"); return
    a, b = marks[-2] + 1, marks[-1] + 1          # the last complete step: after the previous marker up to and including this one
    st = rows[a:b]
    t0, t1 = st[0][0], max(r[1] for r in st)
    busy, cur_s, cur_e = 0, None, None
    gaps = []
    for s, e, n, q in st:
        if cur_e is None:
            cur_s, cur_e = s, e
        elif s > cur_e:
            busy += cur_e - cur_s; gaps.append((s - cur_e, n)); cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"step: {len(st)} launches, wall {(t1 - t0) / 1e6:.3f} ms, union busy {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms, sum of kernels {sum(e - s for s, e, _, _ in st) / 1e6:.3f} ms")
    agg = {}
    for s, e, n, q in st:
        v = agg.setdefault(n, [0, 0]); v[0] += 1; v[1] += e - s
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"  {t / 1e6:8.3f} ms  x{c:4d}  {t / c / 1e3:8.2f} us  {n}")
    gaps.sort(reverse=True)
    print("largest idle gaps (us, before kernel):", [(round(g / 1e3, 1), n[:40]) for g, n in gaps[:12]])
    print(f"gaps > 1 us: {sum(1 for g, _ in gaps if g > 1000)}, total {sum(g for g, _ in gaps if g > 1000) / 1e6:.3f} ms")
    if seq:
        prev = t0
        for s, e, n, q in st:
            print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f}  q{q}  {n}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "adam_kernel", "--seq" in sys.argv)
