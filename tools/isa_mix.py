#!/usr/bin/env python3
"""Static instruction mix of the kernels in a hipcc -S listing: tools/isa_mix.py file.s [name-substring].
Counts are per kernel over the whole text (loop bodies once): a first look at whether a kernel's inner loop is issue bound
on the VALU, on the matrix cores or neither -- not a replacement for a trace."""
import collections
import re
import sys


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    name = None
    counts = {}
    for line in text.split("\n"):
        m = re.match(r"^(_Z[\w]+):", line)
        if m:
            name = m.group(1)
            counts[name] = collections.Counter()
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            name = None
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s", line)
        if not m:
            continue
        op = m.group(1)
        c = counts[name]
        if op.startswith("v_mfma"):
            c["mfma " + op[7:]] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
        elif op.startswith("s_waitcnt"):
            c["waitcnt"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    for k, c in counts.items():
        if want in k and c:
            print(k[:110])
            print("   ", dict(c))


if __name__ == "__main__":
    main()
