"""No kernel of the library may spill vector registers: at the end of round 2 an instantiation of the probe kernel that spilled 46 VGPRs returned
wrong values out of a reload for one register assignment.  The register budgets (`__launch_bounds__`) are chosen so that nothing
spills; this reads the code object's own metadata (`.vgpr_spill_count`, `.sgpr_spill_count`, `.private_segment_fixed_size` of every kernel) back out of the built library."""
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
    blocks = re.split(r"\n\s+- \.agpr_count", notes)[1:]
    assert len(blocks) > 100
    bad = []
    for b in blocks:
        name = re.search(r"\.name:\s+(\S+)", b).group(1)
        g = lambda f: int(re.search(r"\." + f + r":\s+(\d+)", b).group(1))
        # (1) no vector register may be spilled: that is the configuration that returned wrong values
        if g("vgpr_spill_count"):
            bad.append(f"{name}: {g('vgpr_spill_count')} VGPRs spilled")
        # (2) scalar registers ARE spilled by the probe kernels (30-65 of them: kernel arguments and loop state) -- into the lanes of spare VGPRs, which is
        # safe exactly as long as such a VGPR never goes to scratch memory itself (an EXEC-masked scratch store would drop the lanes that hold another
        # thread-independent value): a kernel that spills SGPRs must therefore not use scratch at all
        if g("sgpr_spill_count") and g("private_segment_fixed_size"):
            bad.append(f"{name}: {g('sgpr_spill_count')} SGPRs spilled to VGPR lanes AND {g('private_segment_fixed_size')} bytes of scratch")
    assert not bad, "register budgets (`__launch_bounds__`) must leave every kernel free of vector spills:\n" + "\n".join(bad)
