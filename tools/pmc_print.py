#!/usr/bin/env python3
"""prints the per-kernel averages of the rocprofv3 --pmc passes tools/pmc_kernel.sh wrote under <dir> (rocpd sqlite output)"""
import glob
import os
import sqlite3
import sys

for path in sorted(glob.glob(os.path.join(sys.argv[1], "*", "*.db"))):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    T = lambda n: [t for t in tabs if t.startswith(n)][0]  # noqa: E731
    q = ("select ks.kernel_name, ip.name, avg(pe.value), count(*) from %s pe join %s ip on pe.pmc_id=ip.id join %s kd on pe.event_id=kd.event_id "
         "join %s ks on kd.kernel_id=ks.id group by 1,2" % (T("rocpd_pmc_event"), T("rocpd_info_pmc"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")))
    for name, ctr, val, n in c.execute(q):
        if "reduce" in name or "repack" in name:
            continue
        print("%-60s %-32s %16.0f n=%d" % (name.split("(")[0][-60:], ctr, val, n))
