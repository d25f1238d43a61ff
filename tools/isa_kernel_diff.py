#!/usr/bin/env python3
"""Which kernels of two `hipcc -S --cuda-device-only` listings of one .hip file differ?  (labels normalised, comments dropped)
A change meant for one instantiation of a shared template body must leave the others byte-identical before its A/B means anything
(method note).  usage: isa_kernel_diff.py base.s new.s"""
import re, sys


def kernels(fn):
    t = open(fn).read().split("\n")
    out, name, body = {}, None, []
    for l in t:
        m = re.match(r"^(_Z\S+):\s*;?\s*@", l) or re.match(r"^(_Z\S+):\s*$", l)
        if m and not l.startswith("."):
            name, body = m.group(1), []
            continue
        if l.startswith(".Lfunc_end") and name:
            out[name] = body
            name = None
            continue
        if name is not None and not l.strip().startswith(";"):
            body.append(re.sub(r"\.LBB\d+_\d+", "L", l.split(";")[0].rstrip()))
    return out


a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
for k in sorted(set(a) | set(b)):
    if k not in a or k not in b:
        print(f"{k}: only in {'base' if k in a else 'new'}")
    else:
        print(f"{k}: {len(a[k])} / {len(b[k])} lines: {'identical' if a[k] == b[k] else 'DIFFERENT'}")
