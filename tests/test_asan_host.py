"""api.cpp's HOST side under AddressSanitizer + LeakSanitizer on a HIP stub (tests/hipstub/): the create / load / finalize /
bank / clone / decode-graph / stream / destroy orders that crashed an in-process loop on the GPU in round 1, plus the ViECap
entry points, twelve iterations.  GPU AddressSanitizer is not available on the pool; kernels do not run here -- this is
about ownership and lifetimes in the C ABI (handles, clones borrowing weights, graph caches, event pools, staging slots)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CLANGXX = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(CLANGXX)), reason="needs the ROCm clang toolchain")
def test_c_abi_host_side_is_clean_under_asan(tmp_path):
    flags = ["-x", "hip", "--cuda-host-only", "-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer", "-std=c++17",
             "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "patchioner_amd", "csrc")]
    objs = []
    for src in (os.path.join(ROOT, "patchioner_amd", "csrc", "api.cpp"), os.path.join(ROOT, "tests", "hipstub", "hip_stub.cpp"),
                os.path.join(ROOT, "tests", "hipstub", "asan_loop.cpp")):
        obj = str(tmp_path / (os.path.basename(src) + ".o"))
        subprocess.run([HIPCC] + flags + ["-c", src, "-o", obj], check=True, capture_output=True)
        objs.append(obj)
    exe = str(tmp_path / "asan_loop")
    subprocess.run([CLANGXX, "-fsanitize=address"] + objs + ["-o", exe], check=True, capture_output=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "asan loop ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-4000:]
