"""Mean of every counter of a rocprofv3 --pmc counter_collection.csv per (kernel, grid size), with two derived figures
for MFMA kernels on gfx950 (MI355X_MICROARCH.md, cycle constants):

    clock_ghz  = GRBM_GUI_ACTIVE / 8 / duration           (the counter is summed over the 8 XCDs)
    mfma_util  = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8)

    python tools/pmc_counters.py <counter_collection.csv> [<kernel_trace.csv>] [substring filter] [out.json]
"""
import collections
import csv
import json
import sys


def short(name):
    return name.replace("void ", "").split("(")[0][:90]


def main():
    path = sys.argv[1]
    trace = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2].endswith(".csv") else None
    rest = [a for a in sys.argv[2:] if a != trace]
    flt = rest[0] if rest and not rest[0].endswith(".json") else ""
    out_json = next((a for a in rest if a.endswith(".json")), None)
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    with open(path) as f:
        for r in csv.DictReader(f):
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
            if flt and flt not in key[0]:
                continue
            c = acc[key][r["Counter_Name"]]
            c[0] += 1
            c[1] += float(r["Counter_Value"])
            if not trace and "Start_Timestamp" in r:        # the dispatch's own bracket (inflated by the collection itself)
                d = acc[key]["duration_us_profiled"]
                d[0] += 1
                d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    dur = collections.defaultdict(lambda: [0, 0.0])
    if trace:
        with open(trace) as f:
            for r in csv.DictReader(f):
                key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
                d = dur[key]
                d[0] += 1
                d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    out = {}
    for key in sorted(acc):
        m = {k: v[1] / v[0] for k, v in acc[key].items()}
        m["launches"] = max(v[0] for v in acc[key].values())
        if key in dur:
            m["duration_us"] = dur[key][1] / dur[key][0]
        if "GRBM_GUI_ACTIVE" in m:
            act = m["GRBM_GUI_ACTIVE"] / 8.0
            du = m.get("duration_us", m.get("duration_us_profiled"))
            if du:
                m["clock_ghz"] = act / du * 1e-3
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                m["mfma_util"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * 256.0 * act)
        if "SQ_WAVE_CYCLES" in m:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if k in m:
                    m[k + "_frac"] = m[k] / m["SQ_WAVE_CYCLES"]
        if "SQ_LDS_IDX_ACTIVE" in m and "SQ_LDS_BANK_CONFLICT" in m and m["SQ_LDS_IDX_ACTIVE"] > 0:
            m["lds_conflict_frac"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
        out["%s grid %d" % key] = m
        print("%s grid %d" % key)
        print("   " + "  ".join("%s=%.4g" % (k, v) for k, v in sorted(m.items())))
    if out_json:
        json.dump(out, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
