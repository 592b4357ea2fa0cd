#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV into the short, committed
summary under profiles/ (one line per kernel, names shortened)."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(aof::[A-Za-z0-9_]+(<[^>]*>)?)", name)
    if m:
        return m.group(1)
    return (name[:70] + "...") if len(name) > 73 else name


def main():
    path, title = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    rows = list(csv.DictReader(open(path)))
    print(f"# rocprofv3 --kernel-trace --stats summary: {title}")
    print(f"# source: {path}")
    print(f"{'kernel':<60} {'calls':>6} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'total_ms':>10} {'pct':>6}")
    for r in rows:
        print(f"{short(r['Name']):<60} {r['Calls']:>6} {float(r['AverageNs'])/1e3:>10.2f} "
              f"{float(r['MinNs'])/1e3:>10.2f} {float(r['MaxNs'])/1e3:>10.2f} "
              f"{float(r['TotalDurationNs'])/1e6:>10.3f} {float(r['Percentage']):>6.2f}")


if __name__ == "__main__":
    main()
