#!/usr/bin/env python
"""Average PMC counter values per dispatch of the decode/encode kernels from rocprofv3 rocpd dbs."""
import glob
import os
import sqlite3
import sys


def main(root):
    for db in sorted(glob.glob(os.path.join(root, "p*", "**", "*.db"), recursive=True)):
        con = sqlite3.connect(db)
        try:
            cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
        except Exception as e:
            print(db, "no counters_collection", e); continue
        if not cols:
            print(db, "empty"); continue
        # find name columns
        namecol = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else cols[0])
        cn = "counter_name" if "counter_name" in cols else None
        cv = "value" if "value" in cols else ("counter_value" if "counter_value" in cols else None)
        if not cn or not cv:
            print(db, cols); continue
        q = f"select {namecol}, {cn}, count(*), sum({cv}) from counters_collection group by {namecol}, {cn}"
        ndisp = {}
        for kname, counter, n, tot in con.execute(q):
            if "zn_k" not in kname:
                continue
            short = kname.split("(")[0][-40:]
            disp = con.execute(f"select count(distinct dispatch_id) from counters_collection where {namecol}=? and {cn}=?", (kname, counter)).fetchone()[0] if "dispatch_id" in cols else n
            print(f"{short:40s} {counter:24s} per-dispatch {tot / max(disp, 1):16.1f}  (dispatches {disp})")


if __name__ == "__main__":
    main(sys.argv[1])
