#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) as a per-kernel table (markdown)."""
import sqlite3
import sys


def main(db, title):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"# {title}\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, n, s, a, mn, mx in rows:
        print(f"| `{name[:110]}` | {n} | {s/1e6:.3f} | {a/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*s/tot:.1f} |")
    extra = [x for x in ("vgpr_count", "accum_vgpr_count", "sgpr_count", "lds_size", "scratch_size") if x in cols]
    if extra:
        print("\n| kernel | " + " | ".join(extra) + " |\n|---|" + "---:|" * len(extra))
        for r in c.execute(f"select name, {', '.join('max(' + x + ')' for x in extra)} from kernels group by name order by sum(end-start) desc"):
            print(f"| `{r[0][:110]}` | " + " | ".join(str(x) for x in r[1:]) + " |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "rocprofv3 kernel stats")
