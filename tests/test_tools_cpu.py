"""The profile post-processing tools (tools/pmc_summary.py, tools/trace_by_grid.py) on tiny synthetic rocprofv3 CSVs:
bench.py's `roofline.traffic` and the per-launch-shape durations in profiles/ come out of them."""
import csv
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEMM = "_ZN3pio10k_vit_gemmIDF16_Li1ELi128ELi1EEEvNS_8GemmArgsE"


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _counter_csv(path, counter, rows):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Grid_Size", "Kernel_Name", "Counter_Name", "Counter_Value"])
        for i, (name, grid, value) in enumerate(rows):
            w.writerow([i, grid, name, counter, value])


def test_pmc_summary_separates_shared_launches_by_grid(tmp_path):
    pm = _load("pmc_summary")
    # synchronous run: 16-image launches only (grid 152064); pipelined run: the same plus 64-image launches (grid 608256)
    _counter_csv(tmp_path / "sf.csv", "FETCH_SIZE", [(GEMM, 152064, 1000.0)] * 4 + [("other_kernel", 64, 5.0)])
    _counter_csv(tmp_path / "sw.csv", "WRITE_SIZE", [(GEMM, 152064, 500.0)] * 4)
    _counter_csv(tmp_path / "pf.csv", "FETCH_SIZE", [(GEMM, 152064, 1000.0)] * 2 + [(GEMM, 608256, 4000.0)] * 3)
    _counter_csv(tmp_path / "pw.csv", "WRITE_SIZE", [(GEMM, 152064, 500.0)] * 2 + [(GEMM, 608256, 2000.0)] * 3)
    avg, detail = pm.shared_launch_gemm(str(tmp_path / "pf.csv"), str(tmp_path / "pw.csv"), str(tmp_path / "sf.csv"))
    assert list(detail) == ["%s grid 608256" % GEMM.split("(")[0][:80]]
    assert detail[list(detail)[0]]["launches"] == 3
    assert avg == (2.0 * 4000.0 + 2000.0) * 1024.0          # FETCH_SIZE doubled (gfx950 correction), KiB -> bytes
    out = tmp_path / "traffic.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), str(tmp_path / "sf.csv"), str(tmp_path / "sw.csv"),
                    str(out), str(tmp_path / "pf.csv"), str(tmp_path / "pw.csv")], check=True, capture_output=True)
    tj = json.loads(out.read_text())
    assert tj["vit_gemm_hbm_bytes_per_launch"] == (2.0 * 1000.0 + 500.0) * 1024.0
    assert tj["vit_gemm_hbm_bytes_per_launch_pipelined"] == avg


def test_trace_by_grid_splits_launch_shapes(tmp_path):
    path = tmp_path / "trace.csv"
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"])
        for i in range(3):
            w.writerow([GEMM, 1000 * i, 1000 * i + 30, 152064, 1, 1])
        for i in range(2):
            w.writerow([GEMM, 5000 + 1000 * i, 5000 + 1000 * i + 90, 608256, 1, 1])
        w.writerow(["pio::k_layernorm", 0, 10, 64, 1, 1])
    out = tmp_path / "by_grid.csv"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_by_grid.py"), str(path), "k_vit_gemm", str(out)], check=True)
    rows = list(csv.DictReader(open(out)))
    assert [(r["Grid_Size"], r["Calls"], r["AverageNs"]) for r in rows] == [("152064", "3", "30.0"), ("608256", "2", "90.0")]
