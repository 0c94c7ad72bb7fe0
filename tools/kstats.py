import sqlite3, sys, glob
for f in sys.argv[1:]:
    c=sqlite3.connect(f)
    tabs=[r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    kd=[t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]; ks=[t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    print(f)
    for r in c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by 1 order by 3 desc limit 8"):
        print("  ", r[0][:50].ljust(50), r[1], round(r[2],2), round(r[3],1))
