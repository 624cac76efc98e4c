"""The schedule packers of librtc_hip (csrc/rtc_schedule.h) on the CPU: tools/sanitize/pack_fuzz.hip feeds them random
per-pixel costs, per-chunk times, rectangle and tile pixel maps and 4 / 256 / 2048 waves; every packed schedule must hand
out every pixel exactly once and chunkTimes must return all of the measured time.  The HIP sources are compiled
--cuda-host-only (no kernel is launched); tools/sanitize_host.sh runs the same driver under ASan + UBSan."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(CLANG)), reason="needs the ROCm compilers")
def test_packers_hand_out_every_pixel_once(tmp_path):
    flags = ["--offload-arch=gfx950", "--cuda-host-only", "-std=c++17", "-O1", "-ffp-contract=off", "-fPIC",
             f"-I{REPO}/include"]
    objs = []
    for name, src in (("pack_fuzz", f"{REPO}/tools/sanitize/pack_fuzz.hip"),
                      ("rtc_kernels", f"{REPO}/ray-tracer-challenge_amd/csrc/rtc_kernels.hip")):
        obj = str(tmp_path / f"{name}_host.o")
        subprocess.run([HIPCC, *flags, "-c", "-o", obj, src], check=True, capture_output=True, timeout=600)
        objs.append(obj)
    # the host objects expect their device code under these symbols; none is launched here
    syms = set()
    for obj in objs:
        syms.update(m.decode() for m in re.findall(rb"__hip_fatbin_[0-9a-f]+", open(obj, "rb").read()))
    stub = tmp_path / "no_device_code.cpp"
    stub.write_text("".join(f'extern "C" const char {s}[64] __attribute__((aligned(4096))) = {{0}};\n' for s in sorted(syms)))
    exe = str(tmp_path / "pack_fuzz")
    subprocess.run([CLANG, "-std=c++17", "-o", exe, str(stub), *objs, "-L/opt/rocm/lib", "-lamdhip64",
                    "-Wl,-rpath,/opt/rocm/lib"], check=True, capture_output=True, timeout=600)
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    m = re.search(r"pack_fuzz: (\d+) cases, (\d+) failures", run.stdout)
    assert m and int(m.group(1)) > 100 and int(m.group(2)) == 0, run.stdout[-2000:]
