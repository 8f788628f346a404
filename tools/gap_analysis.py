"""GPU busy fraction of the steady-state part of a rocprofv3 kernel trace (rocpd sqlite): sum of kernel durations / wall span,
and the distribution of gaps between consecutive kernels.  python tools/gap_analysis.py <results.db> [lo hi]  (window of the trace, fractions of its span)"""
import sqlite3
import sys


def main(db_path, lo=0.45, hi=0.7):
    db = sqlite3.connect(db_path)
    rows = db.execute("select start, end, name from kernels order by start").fetchall()
    t0, t1 = rows[0][0], rows[-1][1]
    rows = [r for r in rows if t0 + (t1 - t0) * lo <= r[0] <= t0 + (t1 - t0) * hi]
    busy = sum(e - s for s, e, _ in rows)
    span = rows[-1][1] - rows[0][0]
    gaps = sorted(max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1))
    n = len(gaps)
    by = {}
    for st, en, name in rows:
        a = by.setdefault(name, [0, 0])
        a[0] += 1; a[1] += en - st
    for name, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:22]:
        print(f"{100.0 * t / busy:6.2f} %  calls {c:6d}  avg {t / c / 1e3:8.1f} us  {name[:100]}")
    big = [g for g in gaps if g > 20000]
    print(f"gaps > 20 us: {len(big)} totalling {sum(big) / 1e6:.2f} ms")
    # which kernel pairs sit either side of the long gaps (host synchronisation points show up here)
    where = {}
    for i in range(len(rows) - 1):
        g = rows[i + 1][0] - rows[i][1]
        if g > 20000:
            k = (rows[i][2][:48], rows[i + 1][2][:48])
            a = where.setdefault(k, [0, 0]); a[0] += 1; a[1] += g
    for (a_, b_), (c, t) in sorted(where.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"   {c:4d} gaps {t / 1e6:8.2f} ms  after [{a_}]  before [{b_}]")
    # in-step boundaries (3 - 20 us): which predecessor leaves the longest idle behind it (dirty-line write-back, dispatch ramp of the successor)
    mid = {}
    tot_mid = 0
    for i in range(len(rows) - 1):
        g = rows[i + 1][0] - rows[i][1]
        if 3000 < g <= 20000:
            a = mid.setdefault((rows[i][2][:60], rows[i + 1][2][:40]), [0, 0]); a[0] += 1; a[1] += g; tot_mid += g
    print(f"gaps of 3 - 20 us: {sum(c for c, _ in mid.values())} totalling {tot_mid / 1e6:.2f} ms; gaps <= 3 us total {sum(g for g in gaps if g <= 3000) / 1e6:.2f} ms")
    for (a_, b_), (c, t) in sorted(mid.items(), key=lambda kv: -kv[1][1])[:10]:
        print(f"   {c:4d} gaps avg {t / c / 1e3:6.1f} us  after [{a_}]  before [{b_}]")
    print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms ({100.0 * busy / span:.1f} %)  "
          f"gap total {sum(gaps) / 1e6:.2f} ms  median {gaps[n // 2] / 1e3:.2f} us  p90 {gaps[int(n * 0.9)] / 1e3:.2f} us  max {gaps[-1] / 1e3:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], *(float(v) for v in sys.argv[2:4]))
