"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into the `--stats`-style per-kernel summary CSV."""
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                      "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out_path, "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage\n")
        for n, c, s, a, mn, mx in rows:
            f.write(f"\"{n}\",{c},{s},{a:.1f},{mn},{mx},{100.0 * s / total:.2f}\n")
    for n, c, s, a, mn, mx in rows[:14]:
        print(f"{100.0 * s / total:6.2f}%  calls={c:6d}  avg={a / 1e3:9.1f} us  total={s / 1e6:9.2f} ms  {n[:90]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
