#!/usr/bin/env python
"""Turn rocprofv3 (ROCm 7.2, rocpd sqlite) output into the small tables we commit under profiles/.

    rocprofv3 --kernel-trace --stats -d DIR -o NAME -- python bench.py ...
    python profiles/summarize_rocpd.py kernels DIR/NAME_results.db        > profiles/rNN_kernel_stats.md
    rocprofv3 --pmc C1 C2 ... -d DIR -o NAME -- python bench.py ...
    python profiles/summarize_rocpd.py pmc DIR/NAME_results.db [filter]   > profiles/rNN_pmc.md

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


if __name__ == "__main__":
    mode, path = sys.argv[1], sys.argv[2]
    if mode == "kernels":
        kernels(path)
    else:
        pmc(path, sys.argv[3] if len(sys.argv) > 3 else "")
