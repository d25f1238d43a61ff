#!/usr/bin/env python3
"""bin2inc.py in.co out.inc — a code object as a C byte array named after the file (attn_fwd_d128.co -> kAttnFwdD128Co)."""
import os, sys
src, dst = sys.argv[1], sys.argv[2]
data = open(src, "rb").read()
stem = os.path.splitext(os.path.basename(src))[0]
name = "k" + "".join(p.capitalize() for p in stem.split("_")) + "Co"
with open(dst, "w") as f:
    f.write(f"// generated from {os.path.basename(src)} ({len(data)} bytes)\nalignas(4096) static const unsigned char {name}[] = {{\n")
    for i in range(0, len(data), 24):
        f.write(" " + ",".join(str(b) for b in data[i:i + 24]) + ",\n")
    f.write("};\n")
