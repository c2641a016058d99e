#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (default output of `rocprofv3 --kernel-trace --stats`) into the small CSV
summaries kept under profiles/.  usage: prof_summary.py <results.db> <out_prefix>"""
import sqlite3
import sys


def main(db_path, prefix):
    cur = sqlite3.connect(db_path).cursor()
    with open(prefix + "_kernel_stats.csv", "w") as f:
        f.write("name,calls,total_us,average_us,percentage\n")
        for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            f.write(",".join(str(x) for x in r) + "\n")
    with open(prefix + "_dispatches.csv", "w") as f:
        f.write("name,duration_ns,grid_x,workgroup_x,lds_bytes,vgpr,accum_vgpr,sgpr,scratch\n")
        q = ("select name,duration,grid_x,workgroup_x,lds_size,vgpr_count,accum_vgpr_count,sgpr_count,scratch_size "
             "from kernels where name not like '__amd%' order by start")
        for r in cur.execute(q):
            f.write(",".join(str(x) for x in r) + "\n")
    try:
        rows = list(cur.execute("select * from counters_collection limit 100000"))
        if rows:
            cols = [d[0] for d in cur.description]
            with open(prefix + "_counters.csv", "w") as f:
                f.write(",".join(cols) + "\n")
                for r in rows:
                    f.write(",".join(str(x) for x in r) + "\n")
    except sqlite3.Error:
        pass


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
