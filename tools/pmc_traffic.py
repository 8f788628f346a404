"""Per-kernel HBM-side traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, separate runs as the
MI355X guide prescribes; units are KiB).  gfx950 correction: FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md, HBM section) => bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  WRITE_SIZE is uncalibrated."""
import json
import re
import sqlite3
import sys


CLASSES = {"xattn_fused": lambda n: "gemm16_kernel<0, 5," in n,                                    # EPI_XATTN: to_q + cross-attention in one launch
           "gemm_dense": lambda n: ("gemm" in n and ("<0," in n) and "gemm16_kernel<0, 5," not in n) or "gemm16_dual_kernel" in n,   # (dual: the grouped Q|K + V^T launch)
           "gemm_conv": lambda n: ("gemm" in n and "<0," not in n and "gemm16_dual_kernel" not in n) or "conv3p" in n,
           "attn_self": lambda n: bool(re.search(r"attn_kernel<\d+, \d+, false", n)),       # attn_kernel<DP, KT, CROSS, ...>
           "attn_cross": lambda n: bool(re.search(r"attn_kernel<\d+, \d+, true", n)) or "cross77_kernel" in n}      # round 5: cross77_kernel (xblock.hip)
BENCH_CLASS = {"xattn_fused": "gemm16_kernel<EPI_XATTN> (to_q + cross-attention)", "gemm_dense": "gemm_kernel<A_DENSE>", "gemm_conv": "gemm_kernel<A_CONV3*>", "attn_self": "attn_kernel<self>", "attn_cross": "attn_kernel<cross>"}


def per_kernel(db_path, counter, keep_last=None):
    """keep_last: {class: L} - keep only the LAST L dispatches of each kernel class (the two event-profiled steps of
    `bench.py --roofline-only` are the last launches of the run), so counters and flops_per_launch describe the same launches."""
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, dispatch_id, sum(value), max(duration) from counters_collection where counter_name=? "
                      "group by kernel_name, dispatch_id order by dispatch_id", (counter,)).fetchall()
    if keep_last:
        by_cls = {}
        for r in rows:
            for c, pred in CLASSES.items():
                if pred(r[0]):
                    by_cls.setdefault(c, []).append(r)
                    break
        rows = [r for c, rs in by_cls.items() for r in (rs[-keep_last[c]:] if c in keep_last else rs)]
    agg = {}
    for name, _, v, dur in rows:
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1; a[1] += v; a[2] += dur or 0
    return agg


def main(fetch_db, write_db, out, bench_json=None):
    keep = None
    if bench_json:
        line = json.loads([l for l in open(bench_json) if l.startswith("{")][-1])
        keep = {c: line["roofline"]["per_kernel"][b]["launches"] for c, b in BENCH_CLASS.items() if b in line["roofline"]["per_kernel"]}
    f, w = per_kernel(fetch_db, "FETCH_SIZE", keep), per_kernel(write_db, "WRITE_SIZE", keep)
    res = {}
    for name in sorted(set(f) | set(w), key=lambda n: -(f.get(n, [0, 0, 0])[2])):
        fn, fv, fd = f.get(name, [0, 0.0, 0.0])
        wn, wv, wd = w.get(name, [0, 0.0, 0.0])
        if fn == 0 or wn == 0:
            continue
        res[name] = dict(launches=fn, avg_us=fd / fn / 1e3, fetch_kib_per_launch=fv / fn, write_kib_per_launch=wv / wn,
                         hbm_bytes_per_launch=(2 * fv / fn + wv / wn) * 1024)
    classes = CLASSES
    summary = {}
    for c, pred in classes.items():
        ks = [v for n, v in res.items() if pred(n)]
        nl = sum(k["launches"] for k in ks)
        if nl:
            summary[c] = dict(launches=nl, hbm_bytes_per_launch=sum(k["hbm_bytes_per_launch"] * k["launches"] for k in ks) / nl,
                              avg_us=sum(k["avg_us"] * k["launches"] for k in ks) / nl)
    json.dump(dict(note=__doc__, launches_kept=keep, classes=summary, kernels=res), open(out, "w"), indent=1)
    for c, v in summary.items():
        print(f"{c:12s} launches={v['launches']:6d} avg={v['avg_us']:8.1f} us  HBM-side bytes/launch={v['hbm_bytes_per_launch'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main(*sys.argv[1:5])
