#!/usr/bin/env python3
"""Per-basic-block-region scratch statistics of one function in a device assembly dump:
   python tools/isa_segments.py file.s <mangled-name-prefix>"""
import re
import sys


def main():
    path, name = sys.argv[1], sys.argv[2]
    lines, on = [], False
    for l in open(path):
        if l.startswith(name):
            on = True
        if on:
            lines.append(l.rstrip("\n"))
            if l.startswith(".Lfunc_end"):
                break
    marks = [i for i, l in enumerate(lines) if re.match(r"^\.LBB|\s+s_cbranch", l)]

    def cnt(a, b):
        d = dict(ins=0, ld=0, st=0, bytes_ld=0, bytes_st=0, calls=0, lds=0, waits=0)
        for l in lines[a:b]:
            s = l.strip()
            if not s or s.startswith(";") or s.startswith("."):
                continue
            d["ins"] += 1
            m = re.match(r"(scratch|flat)_(load|store)_dword(x(\d))?", s)
            if m:
                n = int(m.group(4) or 1)
                if m.group(2) == "load":
                    d["ld"] += 1; d["bytes_ld"] += 4 * n
                else:
                    d["st"] += 1; d["bytes_st"] += 4 * n
            if s.startswith("s_swappc"):
                d["calls"] += 1
            if s.startswith("ds_"):
                d["lds"] += 1
            if s.startswith("s_waitcnt"):
                d["waits"] += 1
        return d
    prev = 0
    for m in marks + [len(lines)]:
        if m - prev > 50:
            print(prev, m, cnt(prev, m))
        prev = m


if __name__ == "__main__":
    main()
