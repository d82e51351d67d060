"""Kernel timeline of one bench step from a rocprofv3 rocpd database: start/end (us, relative), queue, short name.
Usage: rocpd_timeline.py file.db [first_kernel_substring] [count] [queue | -] [back]
back: start at the back-th occurrence of the anchor kernel counted from the end (default 4), e.g. to show a step of the timed two-stream region
rather than of the serial pass that follows it."""
import re
import sqlite3
import sys


def main(db_path, anchor="nl_setup", count=70, queue=None, back=4):
    queue = None if queue in (None, "-") else queue
    back = int(back)
    db = sqlite3.connect(db_path)
    rows = db.execute("select d.start, d.end, d.queue_id, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                      "on d.kernel_id = s.id order by d.start").fetchall()
    # start at the LAST-but-some occurrence of the anchor kernel so that a steady-state step is shown
    idx = [k for k, r in enumerate(rows) if anchor in r[3]]
    k0 = idx[-back] if len(idx) >= back else (idx[0] if idx else 0)
    t0 = rows[k0][0]
    shown = 0
    for st, en, q, name in rows[k0:]:
        if queue is not None and str(q) != str(queue):
            continue
        shown += 1
        if shown > int(count):
            break
        m = re.search(r"(\w+)(<[^>]*>)?\(", name.replace("(anonymous namespace)::", ""))
        short = (m.group(1) if m else name)[:38]
        print(f"{(st - t0) / 1e3:9.1f} {(en - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f} us  q{q}  {short}")


if __name__ == "__main__":
    main(*sys.argv[1:])
