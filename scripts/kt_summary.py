"""Per-kernel durations of a rocprofv3 --kernel-trace run (rocpd sqlite output): python scripts/kt_summary.py <dir-or-db> [...]"""
import collections, glob, os, sqlite3, sys


def summarise(path):
    dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
    for dbp in dbs:
        cur = sqlite3.connect(dbp).cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        kd = [t for t in tabs if "kernel_dispatch" in t][0]
        sym = [t for t in tabs if "kernel_symbol" in t][0]
        rows = list(cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {sym} s on d.kernel_id = s.id order by d.start"))
        acc = collections.OrderedDict()
        for n, s, e in rows:
            acc.setdefault(n.split("(")[0][:70], []).append(e - s)
        print(dbp)
        for n, d in acc.items():
            d2 = sorted(d)
            print(f"  {n:70s} calls {len(d):4d}  median {d2[len(d2) // 2] / 1e3:9.1f} us  min {d2[0] / 1e3:9.1f}  mean {sum(d) / len(d) / 1e3:9.1f}")
        last = [r for r in rows if "rtus" in r[0]][-4:]
        if last:
            t0 = last[0][1]
            for n, s, e in last:
                print(f"      last pass: {n.split('(')[0][:44]:44s} start {(s - t0) / 1e3:8.1f} end {(e - t0) / 1e3:8.1f}")


for p in sys.argv[1:]:
    summarise(p)
