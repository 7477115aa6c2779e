#!/usr/bin/env python
"""Per-kernel summary (count / total / avg / min / max, share) of a rocprofv3 rocpd database or
kernel-trace CSV; writes the markdown table committed under profiles/."""
import csv
import sqlite3
import sys


def rows_from(path):
    if path.endswith(".db"):
        c = sqlite3.connect(path)
        return c.execute("select name, end-start from kernels").fetchall()
    out = []
    for r in csv.DictReader(open(path)):
        out.append((r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def main():
    path = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else path
    agg = {}
    for name, dur in rows_from(path):
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values())
    print(f"# {title}\n")
    print(f"total kernel time {tot / 1e6:.3f} ms over {sum(a[0] for a in agg.values())} dispatches\n")
    print("| kernel | calls | total ms | avg ms | min ms | max ms | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) < 90 else name[:87] + "..."
        print(f"| `{short}` | {a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e6:.4f} | {a[2] / 1e6:.4f} | "
              f"{a[3] / 1e6:.4f} | {100 * a[1] / tot:.1f} |")


if __name__ == "__main__":
    main()
