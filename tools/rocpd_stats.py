#!/usr/bin/env python3
"""Summarises a rocprofv3 rocpd .db (kernel trace) into a per-kernel stats table (name, calls, total/avg ms, %)."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels").fetchall() if "name" in cols else []
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e6
    tot = sum(v[1] for v in agg.values())
    lines = [f"# rocprofv3 --kernel-trace summary of {path}", f"# total kernel time {tot:.2f} ms over {len(rows)} dispatches",
             f"{'kernel':112s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}"]
    for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{k:112s} {n:7d} {ms:10.3f} {ms / n * 1e3:10.1f} {100 * ms / tot:6.2f}")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
