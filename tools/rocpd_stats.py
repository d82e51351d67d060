"""Summarise a rocprofv3 rocpd database (.db) into the per-kernel statistics table (`--stats` equivalent):
name, calls, total/avg/min/max duration (ns), share of GPU time, VGPR/SGPR/LDS.  Usage: rocpd_stats.py file.db [out.csv]"""
import csv
import sqlite3
import sys


def summarise(db_path):
    db = sqlite3.connect(db_path)
    rows = db.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start), "
        "max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.workgroup_size_x), max(d.grid_size_x) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    return [dict(name=r[0], calls=r[1], total_ns=r[2], avg_ns=round(r[3], 1), min_ns=r[4], max_ns=r[5], pct=round(100.0 * r[2] / total, 2),
                 vgpr=r[6], sgpr=r[7], lds_bytes=r[8], wg_size=r[9], grid_x=r[10]) for r in rows]


if __name__ == "__main__":
    out = summarise(sys.argv[1])
    w = csv.DictWriter(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout, fieldnames=list(out[0].keys()))
    w.writeheader()
    w.writerows(out)
