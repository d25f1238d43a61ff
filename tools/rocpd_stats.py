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
    # GPU-busy time = union of the kernel intervals (kernels of two streams overlap); the rest of the traced span is idle: launch gaps,
    # host-side pauses (model construction before the first step is cut off by starting at the first GEMM launch)
    iv = sorted((s, e) for name, s, e in rows)
    first = min((s for name, s, e in rows if "gemm_pp4" in name), default=iv[0][0] if iv else 0)
    iv = [(s, e) for s, e in iv if s >= first]
    busy, gaps, cur_s, cur_e = 0, [], None, None
    for s, e in iv:
        if cur_e is None: cur_s, cur_e = s, e
        elif s > cur_e:
            busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    if cur_e is not None: busy += cur_e - cur_s
    span = (iv[-1][1] - iv[0][0]) if iv else 0
    small = [g for g in gaps if g < 100e3]           # (< 100 us: kernel-to-kernel hand-over, not a host pause)
    gap_line = (f"# from the first GEMM launch: span {span / 1e6:.1f} ms, GPU busy (union over streams) {busy / 1e6:.1f} ms = {100.0 * busy / max(1, span):.1f} %; "
                f"{len(small)} kernel-to-kernel gaps < 100 us: total {sum(small) / 1e6:.2f} ms, median {sorted(small)[len(small) // 2] / 1e3 if small else 0:.2f} us; "
                f"{len(gaps) - len(small)} longer pauses: {sum(g for g in gaps if g >= 100e3) / 1e6:.1f} ms")
    lines = [f"# rocprofv3 --kernel-trace summary of {path}", f"# total kernel time {tot:.2f} ms over {len(rows)} dispatches", gap_line,
             f"{'kernel':112s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}"]
    for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{k:112s} {n:7d} {ms:10.3f} {ms / n * 1e3:10.1f} {100 * ms / tot:6.2f}")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
