"""Per-kernel counter sums from a rocprofv3 rocpd database collected with --pmc.  Usage: rocpd_pmc.py file.db out.csv"""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(pmc_events)")]
    disp = next((c for c in ("dispatch_id", "event_id", "stack_id") if c in cols), None)
    try:
        rows = db.execute(
            f"select name as kernel, counter_name, count(*), sum(counter_value), avg(end - start), {'count(distinct ' + disp + ')' if disp else '0'} "
            "from pmc_events group by name, counter_name order by 5 desc").fetchall()
    except Exception as exc:  # schema differs: dump it so the query can be fixed
        print("pmc_events columns:", cols, "error:", exc)
        for t in ("rocpd_pmc_event", "rocpd_info_pmc"):
            print(t, [r[1] for r in db.execute(f"pragma table_info({t})")])
        raise
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        # samples = rows of the counter (SQ counters: one per XCD x SE instance and dispatch; TCC-derived: one per dispatch);
        # per_dispatch = sum / samples (kept for the traffic tools); per_launch = sum / number of distinct dispatches
        w.writerow(["kernel", "counter", "samples", "sum", "per_dispatch", "avg_ns", "dispatches", "per_launch"])
        for k, c, n, v, ns, nd in rows:
            w.writerow([k, c, n, v, v / max(n, 1), round(ns or 0, 1), nd, v / nd if nd else ""])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
