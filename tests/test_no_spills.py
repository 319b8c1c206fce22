"""No kernel of the library may spill vector registers: at the end of round 2 an instantiation of the probe kernel that spilled 46 VGPRs returned
wrong values out of a reload for one register assignment.  The register budgets (`__launch_bounds__`) are chosen so that nothing
spills; this reads the code object's own metadata (`.vgpr_spill_count` of every kernel) back out of the built library."""
import glob, os, re, shutil, subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def test_no_kernel_spills_vector_registers(tmp_path):
    lib = os.path.join(ROOT, "sage2_amd", "libsage2ov.so")
    objdump, readelf = os.path.join(LLVM, "llvm-objdump"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(lib) and os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("library or LLVM tools not present")
    work = tmp_path / "libsage2ov.so"; shutil.copy(lib, work)
    subprocess.run([objdump, "--offloading", str(work)], check=True, stdout=subprocess.DEVNULL, cwd=tmp_path)
    cos = [f for f in glob.glob(str(tmp_path / "libsage2ov.so.*")) if "gfx950" in f]
    assert cos, "no gfx950 code object in the library"
    notes = subprocess.run([readelf, "--notes", cos[0]], check=True, stdout=subprocess.PIPE, text=True).stdout
    kernels = re.findall(r"\.name:\s+(\S+)", notes); spills = [int(x) for x in re.findall(r"\.vgpr_spill_count:\s+(\d+)", notes)]
    assert len(spills) > 100 and len(spills) <= len(kernels)
    bad = [s for s in spills if s != 0]
    assert not bad, f"{len(bad)} kernels spill VGPRs: give them a register budget they fit (see k_probe_fast's __launch_bounds__)"
