#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace) into a per-kernel stats table:
calls, total / average / min / max duration.  Usage: rocprof_summary.py results.db [> summary.txt]"""
import re
import sqlite3
import sys


def main(path: str) -> None:
    c = sqlite3.connect(path)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else scols[-1])
    q = (f"select s.{name_col}, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc")
    rows = list(c.execute(q))
    tot = sum(r[2] for r in rows) or 1
    print(f"# source: {path}\n# columns: kernel | calls | total_us | avg_us | min_us | max_us | pct")
    for name, n, total, avg, mn, mx in rows:
        short = re.sub(r"\s+", " ", str(name))[:110]
        print(f"{short:110s} {n:6d} {total/1e3:12.1f} {avg/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100.0*total/tot:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
