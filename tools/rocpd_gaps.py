#!/usr/bin/env python3
"""Busy / idle split of the GPU timeline from a rocprofv3 rocpd database: how much of the wall time between the first
and last dispatch of the steady-state region is spent inside kernels, and how much in the gaps between them.

    python tools/rocpd_gaps.py gpurun_out/prof/run_results.db [lo=0.3] [hi=0.8]     (window, as fractions of all dispatches)
"""
import sqlite3
import sys


def main(path: str, lo: float = 0.3, hi: float = 0.8) -> None:
    db = sqlite3.connect(path)
    rows = db.execute("""select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d
                         join rocpd_info_kernel_symbol s on d.kernel_id = s.id and d.guid = s.guid order by d.start""").fetchall()
    rows = rows[int(len(rows) * lo):int(len(rows) * hi)]
    wall = rows[-1][1] - rows[0][0]
    busy, gaps, last_end = 0, [], rows[0][0]
    for st, en, _ in rows:
        if st > last_end:
            gaps.append(st - last_end)
        busy += max(0, en - max(st, last_end))
        last_end = max(last_end, en)
    gaps.sort()
    print(f"# {path}: dispatches in the [{lo}, {hi}] window: {len(rows)}")
    print(f"wall {wall / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms ({100 * busy / wall:.1f}%), idle {(wall - busy) / 1e6:.3f} ms in {len(gaps)} gaps")
    if gaps:
        q = lambda p: gaps[min(len(gaps) - 1, int(p * len(gaps)))] / 1e3
        print(f"gap us: median {q(0.5):.2f}, p90 {q(0.9):.2f}, p99 {q(0.99):.2f}, max {gaps[-1] / 1e3:.2f}; sum of gaps < 20us: "
              f"{sum(g for g in gaps if g < 20000) / 1e6:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1], *(float(a) for a in sys.argv[2:4]))
