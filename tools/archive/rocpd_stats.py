#!/usr/bin/env python3
"""Per-kernel summary (count, average, total) of a rocprofv3 rocpd SQLite result file -> CSV on stdout."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    q = ("select s.kernel_name, count(*), avg(d.end - d.start), sum(d.end - d.start), min(d.end - d.start), "
         "max(d.end - d.start) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
         "group by s.kernel_name order by 4 desc")
    print("Name,Calls,AverageNs,TotalDurationNs,MinNs,MaxNs")
    for name, n, avg, tot, mn, mx in cur.execute(q):
        print(f'"{name}",{n},{avg:.0f},{tot},{mn},{mx}')


if __name__ == "__main__":
    main(sys.argv[1])
