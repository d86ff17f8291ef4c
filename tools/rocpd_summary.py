#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace [+ PMC]) as a per-kernel table.

usage: rocpd_summary.py results.db [--pmc]   -> CSV on stdout
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    want_pmc = "--pmc" in sys.argv
    names = dict(db.execute("select id, kernel_name from rocpd_info_kernel_symbol"))
    rows = db.execute("select kernel_id, start, end, id, event_id from rocpd_kernel_dispatch").fetchall()
    agg = {}
    for kid, s, e, did, evid in rows:
        a = agg.setdefault(kid, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    pmc = {}
    if want_pmc:
        pmc_names = dict(db.execute("select id, name from rocpd_info_pmc"))
        ev2k = {evid: kid for kid, _, _, _, evid in rows}
        for evid, pid, val in db.execute("select event_id, pmc_id, value from rocpd_pmc_event"):
            k = ev2k.get(evid)
            if k is None:
                continue
            pmc.setdefault(k, {}).setdefault(pmc_names[pid], []).append(val)
    cols = sorted({c for v in pmc.values() for c in v})
    print(",".join(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"] +
                   [f"{c}_avg_per_launch" for c in cols]))
    for kid, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        name = names.get(kid, str(kid)).replace(",", ";")
        extra = []
        for c in cols:
            v = pmc.get(kid, {}).get(c, [])
            extra.append(f"{sum(v) / len(v):.1f}" if v else "")
        print(",".join([f'"{name}"', str(a[0]), str(a[1]), f"{a[1] / a[0]:.1f}", f"{100.0 * a[1] / total:.2f}", str(a[2]), str(a[3])] + extra))


if __name__ == "__main__":
    main()
