#!/usr/bin/env python
"""Turn rocprofv3 (ROCm 7.2, rocpd sqlite) output into the small tables we commit under profiles/.

    rocprofv3 --kernel-trace --stats -d DIR -o NAME -- python bench.py ...
    python profiles/summarize_rocpd.py kernels DIR/NAME_results.db        > profiles/rNN_kernel_stats.md
    rocprofv3 --pmc C1 C2 ... -d DIR -o NAME -- python bench.py ...
    python profiles/summarize_rocpd.py pmc DIR/NAME_results.db [filter]   > profiles/rNN_pmc.md
    python profiles/summarize_rocpd.py gaps DIR/NAME_results.db           (idle gaps between kernels, steady state)

(The sqlite files are tens of MB; they are summarised on the GPU box and deleted.)
"""
import sqlite3
import sys


def short(name, n=80):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def kernels(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("| kernel | calls | total_us | avg_us | % |")
    print("|---|---:|---:|---:|---:|")
    for name, calls, total, avg, pct in rows:
        print(f"| `{short(name)}` | {calls} | {total:.1f} | {avg:.2f} | {pct:.2f} |")
    tot = sum(r[2] for r in rows)
    ours = sum(r[2] for r in rows if not r[0].startswith(("at::", "void at::", "__amd", "void (anonymous namespace)::elementwise"))
               and "at::native" not in r[0])
    print(f"\nall kernels: {tot:.1f} us; library (non-torch) kernels: {ours:.1f} us; torch/runtime kernels: {tot - ours:.1f} us")


def pmc(path, flt=""):
    cur = sqlite3.connect(path).cursor()
    q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
         "group by kernel_name, counter_name")
    table = {}
    for kname, cname, cnt, val, dur in cur.execute(q):
        if flt and flt not in kname:
            continue
        d = table.setdefault(short(kname, 60), {"_n": cnt, "_dur_us": (dur or 0) / 1e3})
        d[cname] = val
    names = sorted({c for d in table.values() for c in d if not c.startswith("_")})
    print("| kernel | dispatches | avg_us | " + " | ".join(names) + " |")
    print("|---|---:|---:|" + "---:|" * len(names))
    for k, d in sorted(table.items(), key=lambda kv: -kv[1]["_dur_us"]):
        print(f"| `{k}` | {d['_n']} | {d['_dur_us']:.1f} | " + " | ".join(f"{d.get(c, 0):.4g}" for c in names) + " |")




def gaps(path, skip_frac=0.5):
    """Idle gaps between consecutive kernels in the second half of the trace (steady state): where does the GPU wait
    for the host?  Prints busy/span and the largest gaps with the kernels on either side."""
    con = sqlite3.connect(path)
    cur = con.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    cand = [t for t in tables if "kernel_dispatch" in t or t == "kernels"]
    rows = None
    for t in cand:
        cols = [r[1] for r in cur.execute(f"pragma table_info({t})")]
        if "start" in cols and "end" in cols:
            namecol = "name" if "name" in cols else ("kernel_name" if "kernel_name" in cols else None)
            if namecol is None and "kernel_id" in cols:
                sym = [x for x in tables if "kernel_symbol" in x]
                if sym:
                    rows = list(cur.execute(f"select s.kernel_name, d.start, d.end from {t} d join {sym[0]} s on d.kernel_id = s.id order by d.start"))
                    break
            elif namecol:
                rows = list(cur.execute(f"select {namecol}, start, end from {t} order by start"))
                break
    if rows is None:
        print("no dispatch table with start/end found; tables:", tables)
        return
    rows = rows[int(len(rows) * skip_frac):]
    span = rows[-1][2] - rows[0][1]
    busy, cur_end, gl = 0, rows[0][1], []
    for i, (name, st, en) in enumerate(rows):
        if st > cur_end:
            gl.append((st - cur_end, short(rows[i - 1][0], 50), short(name, 50)))
        busy += max(0, en - max(st, cur_end))
        cur_end = max(cur_end, en)
    print(f"dispatches {len(rows)}, span {span / 1e3:.1f} us, busy {busy / 1e3:.1f} us ({100 * busy / span:.1f} %), "
          f"idle {(span - busy) / 1e3:.1f} us in {len(gl)} gaps")
    agg = {}
    for g, a, b in gl:
        k = (a, b)
        agg.setdefault(k, [0, 0])
        agg[k][0] += g
        agg[k][1] += 1
    print("| idle_us total | count | avg_us | after kernel | before kernel |")
    print("|---:|---:|---:|---|---|")
    for (a, b), (tot, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"| {tot / 1e3:.1f} | {cnt} | {tot / cnt / 1e3:.1f} | `{a}` | `{b}` |")


def _dispatch_rows(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    for t in [t for t in tables if "kernel_dispatch" in t or t == "kernels"]:
        cols = [r[1] for r in cur.execute(f"pragma table_info({t})")]
        if "start" in cols and "end" in cols:
            namecol = "name" if "name" in cols else ("kernel_name" if "kernel_name" in cols else None)
            if namecol is None and "kernel_id" in cols:
                sym = [x for x in tables if "kernel_symbol" in x]
                if sym:
                    return list(cur.execute(f"select s.kernel_name, d.start, d.end from {t} d join {sym[0]} s "
                                            f"on d.kernel_id = s.id order by d.start"))
            elif namecol:
                return list(cur.execute(f"select {namecol}, start, end from {t} order by start"))
    return None


def timeline(path, marker="project_fwd", which=-2):
    """One steady-state step, kernel by kernel: `marker` (a substring of the kernel that starts a step) delimits steps;
    prints the `which`-th one with each kernel's duration and the idle gap in front of it."""
    rows = _dispatch_rows(path)
    starts = [i for i, r in enumerate(rows) if marker in r[0]]
    a, b = starts[which], starts[which + 1] if which + 1 < 0 or which + 1 < len(starts) else len(rows)
    t0, prev_end = rows[a][1], rows[a][1]
    print(f"step of {b - a} kernels, {(rows[b - 1][2] - t0) / 1e3:.1f} us")
    print("| t_us | gap_us | dur_us | kernel |")
    print("|---:|---:|---:|---|")
    for name, st, en in rows[a:b]:
        print(f"| {(st - t0) / 1e3:.1f} | {max(0, st - prev_end) / 1e3:.1f} | {(en - st) / 1e3:.1f} | `{short(name, 70)}` |")
        prev_end = max(prev_end, en)


if __name__ == "__main__":
    mode, path = sys.argv[1], sys.argv[2]
    if mode == "kernels":
        kernels(path)
    elif mode == "gaps":
        gaps(path)
    elif mode == "timeline":
        timeline(path, (sys.argv[3:4] or ["project_fwd"])[0], int((sys.argv[4:5] or ["-2"])[0]))
    else:
        pmc(path, sys.argv[3] if len(sys.argv) > 3 else "")
