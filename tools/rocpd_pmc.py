#!/usr/bin/env python3
"""Per-kernel sum / mean of one PMC counter from a rocprofv3 rocpd database.   python tools/rocpd_pmc.py db COUNTER"""
import sqlite3
import sys


def main(path, counter):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    pmc = [t for t in tabs if "pmc" in t.lower()]
    if not any("pmc_event" in t for t in pmc):
        print("tables:", tabs)
        return
    ev = [t for t in pmc if t.startswith("rocpd_pmc_event")][0]
    info = [t for t in pmc if t.startswith("rocpd_info_pmc")][0]
    cols_ev = [r[1] for r in db.execute(f"pragma table_info({ev})")]
    cols_info = [r[1] for r in db.execute(f"pragma table_info({info})")]
    print("#", ev, cols_ev)
    print("#", info, cols_info)
    q = f"""select s.kernel_name, count(*), sum(e.value), avg(e.value) from {ev} e
            join {info} i on e.pmc_id = i.id and e.guid = i.guid
            join rocpd_kernel_dispatch d on e.event_id = d.event_id and e.guid = d.guid
            join rocpd_info_kernel_symbol s on d.kernel_id = s.id and d.guid = s.guid
            where i.name = ? group by s.kernel_name order by 3 desc"""
    rows = db.execute(q, (counter,)).fetchall()
    if len(sys.argv) > 3:                       # also append to a JSON summary: {demangled-ish kernel: {counter: [dispatches, sum]}}
        import json, os, re
        out = json.load(open(sys.argv[3])) if os.path.exists(sys.argv[3]) else {}
        for name, n, tot, avg in rows:
            out.setdefault(name, {})[counter] = [n, tot]
        json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(f"# {counter} per kernel (counter units as reported by rocprofv3)")
    print(f"{'kernel':100s} {'dispatches':>10s} {'sum':>16s} {'mean/dispatch':>16s}")
    for name, n, tot, avg in rows:
        print(f"{name[:100]:100s} {n:10d} {tot:16.1f} {avg:16.2f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
