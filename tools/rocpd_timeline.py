#!/usr/bin/env python3
"""One steady-state train step as a timeline, from a rocprofv3 rocpd database: every dispatch between two consecutive
optimizer kernels with its start (us from the step's first dispatch), duration, hardware queue and how many other
dispatches were running when it started.

    python tools/rocpd_timeline.py gpurun_out/prof/run_results.db [which step, default 8]
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    m = re.search(r"\d+([a-z_0-9]+_kernel)(I.*?E)?Ev", name)
    if m:
        targs = re.findall(r"L[bi](\d+)E", m.group(2) or "")
        return m.group(1) + ("<" + ",".join(targs) + ">" if targs else "")
    return name.split("(")[0][:48]


def main(path: str, which: int = 8) -> None:
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(rocpd_kernel_dispatch)")]
    qcol = "d.queue_id" if "queue_id" in cols else "0"
    rows = db.execute(f"""select d.start, d.end, s.kernel_name, {qcol} from rocpd_kernel_dispatch d
                          join rocpd_info_kernel_symbol s on d.kernel_id = s.id and d.guid = s.guid order by d.start""").fetchall()
    marks = [i for i, r in enumerate(rows) if "adamw_ema" in r[2]]
    lo, hi = marks[which] + 1, marks[which + 1] + 1
    step = rows[lo:hi]
    t0 = step[0][0]
    queues = {q: i for i, q in enumerate(sorted({r[3] for r in step}))}
    print(f"# step {which}: {len(step)} dispatches, {(step[-1][1] - t0) / 1e3:.1f} us, queues {len(queues)}")
    for i, (st, en, name, q) in enumerate(step):
        running = sum(1 for s2, e2, _, _ in step[max(0, i - 12):i] if e2 > st)
        print(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:7.1f}us  q{queues[q]} +{running}  {short(name)}")


if __name__ == "__main__":
    main(sys.argv[1], *(int(a) for a in sys.argv[2:3]))
