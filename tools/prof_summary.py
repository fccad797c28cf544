"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel table:
calls, total ms, average us, share.  Usage: python tools/prof_summary.py results.db [min_share]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    m = re.match(r"(\w+)<(.*)>\(.*", name)
    if m:
        return f"{m.group(1)}<{m.group(2)}>"[:110]
    return name.split("(")[0][:110]


def main(path, min_share=0.002):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for n, s, e in rows:
        k = short(n)
        a = agg.setdefault(k, [0, 0])
        a[0] += 1; a[1] += (e - s)
    tot = sum(v[1] for v in agg.values())
    print(f"total kernel time {tot/1e6:.2f} ms over {len(rows)} dispatches")
    print(f"{'share':>6} {'calls':>7} {'total ms':>10} {'avg us':>10}  kernel")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if t / tot < min_share:
            continue
        print(f"{100*t/tot:6.2f} {c:7d} {t/1e6:10.3f} {t/c/1e3:10.2f}  {k}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.002)
