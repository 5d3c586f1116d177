"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per launch per kernel.

    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> [out.json [<fetch csv> <write csv> of a pipelined run]]

gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts a 128-B request as 64 B for wide coalesced
streaming reads (16 B per lane), so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  Both counters
are in KiB.  Launches of the benchmark's warm-up / timed / profiled regions are averaged together.
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter, by_grid=False):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            key = name.split("(")[0].replace("void ", "")[:80]
            if by_grid:
                key = (key, int(r["Grid_Size"]))
            acc[key][0] += 1
            acc[key][1] += float(r["Counter_Value"])
    return acc


def shared_launch_gemm(fetch_path, write_path, sync_fetch_path):
    """HBM bytes per launch of the k_vit_gemm launches of a pipelined run that do NOT occur in the synchronous run
    (same instantiation and grid size): those are the shared ViT launches (several batches per launch); the others
    belong to the bench's synchronous / profiled regions."""
    fetch = per_kernel(fetch_path, "FETCH_SIZE", True)
    write = per_kernel(write_path, "WRITE_SIZE", True)
    sync = set(per_kernel(sync_fetch_path, "FETCH_SIZE", True))
    tot_b = tot_n = 0.0
    detail = {}
    for key in sorted(fetch):
        if "k_vit_gemm" not in key[0] or key in sync:
            continue
        nf, f = fetch[key]
        nw, w = write.get(key, [0, 0.0])
        b = (2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0
        detail["%s grid %d" % key] = {"launches": nf, "hbm_bytes_per_launch": b}
        tot_b += b * nf
        tot_n += nf
    return tot_b / max(tot_n, 1), detail


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        if "pio" not in k:
            continue
        nf, f = fetch.get(k, [0, 0.0])
        nw, w = write.get(k, [0, 0.0])
        n = max(nf, nw, 1)
        out[k] = {"launches": n, "fetch_kib_raw_per_launch": f / max(nf, 1), "write_kib_per_launch": w / max(nw, 1),
                  "hbm_bytes_per_launch": (2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0}
    gemm = [v for k, v in out.items() if "k_vit_gemm" in k]
    tot_l = sum(v["launches"] for v in gemm)
    import datetime
    summary = {"gemm_kernel": "k_vit_gemm_roll / k_vit_gemm256 / k_vit_gemm",     # bench.py replays these figures only for the kernel named here
               "collected": datetime.date.today().isoformat(),
               "kernels": out,
               "vit_gemm_hbm_bytes_per_launch": sum(v["hbm_bytes_per_launch"] * v["launches"] for v in gemm) / max(tot_l, 1)}
    if len(sys.argv) > 6:      # ... <pmc_counters.py json of an SQ pass over the pipelined run>: MFMA utilisation of the GEMM launches
        pc = json.load(open(sys.argv[6]))
        busy = act = 0.0
        for k, v in pc.items():
            if "k_vit_gemm" in k and "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
                busy += v["SQ_VALU_MFMA_BUSY_CYCLES"] * v["launches"]
                act += v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0 * v["launches"]
        if act > 0:
            summary["mfma_util"] = busy / act
            summary["mfma_util_source"] = ("replayed: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8) over every "
                                           "k_vit_gemm* dispatch of a rocprofv3 --pmc pass of this command (%s)" % sys.argv[6])
    if len(sys.argv) > 5:      # ... <out.json> <fetch csv of the pipelined run> <write csv of the pipelined run>
        avg, detail = shared_launch_gemm(sys.argv[4], sys.argv[5], sys.argv[1])
        summary["vit_gemm_hbm_bytes_per_launch_pipelined"] = avg
        summary["vit_gemm_pipelined_launches"] = detail
        print("vit_gemm, shared launches of the pipelined run: %.2f MB/launch" % (avg / 1e6))
    text = json.dumps(summary, indent=1)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(text)
    for k, v in out.items():
        print("%-70s n=%5d  HBM %.2f MB/launch" % (k, v["launches"], v["hbm_bytes_per_launch"] / 1e6))
    print("vit_gemm average: %.2f MB/launch" % (summary["vit_gemm_hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
