"""Per-kernel SQ counters from one rocprofv3 --pmc pass (rocpd sqlite): MFMA utilisation, LDS bank conflicts, wave stall split.
    MfmaUtil     = SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 256 CUs * 4 SIMDs), cycles = min(GRBM_GUI_ACTIVE / 8 XCDs, 2.4 GHz * duration)
                   (busy cycles are summed over SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs; rows with clock_capped = true had a
                   GUI-active window longer than the dispatch: their utilisation is a lower bound)
    LDS conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE                               (extra cycles / all LDS-array cycles)
    wait split   = SQ_WAIT_ANY, SQ_WAIT_INST_ANY as fractions of SQ_WAVE_CYCLES
python tools/pmc_sq.py <results.db> <out.json>"""
import json
import sqlite3
import sys


def main(db_path, out):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, counter_name, dispatch_id, sum(value), max(duration) from counters_collection "
                      "group by kernel_name, counter_name, dispatch_id").fetchall()
    agg = {}
    for name, ctr, _, v, dur in rows:
        k = agg.setdefault(name, {})
        c = k.setdefault(ctr, [0, 0.0, 0.0])
        c[0] += 1; c[1] += v; c[2] += dur or 0
    res = {}
    for name, k in agg.items():
        g = lambda c: k.get(c, [0, 0.0, 0.0])[1]
        n = max(v[0] for v in k.values())
        dur = max(v[2] for v in k.values())
        gui = g("GRBM_GUI_ACTIVE")
        r = dict(launches=n, avg_us=dur / n / 1e3)
        if gui and dur:
            # GRBM_GUI_ACTIVE spans more than the dispatch for short kernels (ramp-up / drain of the profiled pass): the derived
            # clock then exceeds the part's 2.4 GHz maximum.  Cap the cycle budget at 2.4 GHz x duration and say so.
            cyc = min(gui / 8, 2.4 * dur)
            r["mfma_util"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 256 * 4)
            r["clock_ghz"] = cyc / dur                            # cycles per ns
            r["clock_capped"] = bool(gui / 8 > 2.4 * dur)         # True: utilisation is a lower bound, not a measurement
        if g("SQ_LDS_IDX_ACTIVE"):
            r["lds_conflict_frac"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
        if g("SQ_WAVE_CYCLES"):
            r["wait_any_frac"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
            r["wait_inst_frac"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
        res[name] = r
    top = sorted(res.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches"])
    json.dump(dict(note=__doc__, kernels=dict(top)), open(out, "w"), indent=1)
    for name, r in top[:14]:
        print(f"{r['launches']:5d} x {r['avg_us']:8.1f} us  mfma {100 * r.get('mfma_util', 0):5.1f} %  clk {r.get('clock_ghz') or 0:4.2f} GHz  "
              f"lds-conflict {100 * r.get('lds_conflict_frac', 0):5.1f} %  wait {100 * r.get('wait_any_frac', 0):4.1f} / inst-stall {100 * r.get('wait_inst_frac', 0):4.1f} %  {name[:60]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
