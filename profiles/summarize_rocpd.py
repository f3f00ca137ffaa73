#!/usr/bin/env python
"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into the per-kernel stats table we commit.

    rocprofv3 --kernel-trace --stats -d gpurun_out/profX -o NAME -- python bench.py ...
    python profiles/summarize_rocpd.py gpurun_out/profX/NAME_results.db > profiles/rNN_kernel_stats.md
"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("| kernel | calls | total_us | avg_us | % |")
    print("|---|---:|---:|---:|---:|")
    for name, calls, total, avg, pct in rows:
        name = name.replace("(anonymous namespace)::", "").replace("void ", "")
        if len(name) > 90:
            name = name[:87] + "..."
        print(f"| `{name}` | {calls} | {total:.1f} | {avg:.2f} | {pct:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1])
