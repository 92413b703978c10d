#!/usr/bin/env python3
"""Per-kernel statistics (the `--stats` table) from a rocprofv3 rocpd SQLite database.

    python tools/rocpd_stats.py gpurun_out/prof/run_results.db > profiles/rNN_<what>_kernel_stats.txt
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)          # drop the argument list
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main(path: str) -> None:
    db = sqlite3.connect(path)
    rows = db.execute("""
        select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start),
               max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.group_segment_size)
        from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id and d.guid = s.guid
        group by s.kernel_name order by 3 desc""").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats   (source: {path})")
    print(f"# total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'kernel':110s} {'calls':>7s} {'total_ms':>10s} {'%':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'vgpr':>5s} {'agpr':>5s} {'lds':>6s}")
    for name, n, tot, avg, mn, mx, vg, ag, lds in rows:
        print(f"{short(name):110s} {n:7d} {tot / 1e6:10.3f} {100 * tot / total:6.2f} {avg / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {vg:5d} {ag:5d} {lds:6d}")


if __name__ == "__main__":
    main(sys.argv[1])
