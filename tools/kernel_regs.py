#!/usr/bin/env python3
"""tools/kernel_regs.py <file.s> [pattern] -- VGPRs / SGPRs / spills / LDS / scratch of every kernel whose (demangled) name contains `pattern`, from the
code-object metadata in the assembly that `hipcc -save-temps=obj` leaves behind (sage2ov_device-hip-amdgcn-amd-amdhsa-gfx950.s)."""
import re, subprocess, sys
s = open(sys.argv[1]).read(); pat = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"\n\s+- \.agpr_count", s)[1:]:
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    dn = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
    if pat not in dn: continue
    g = lambda f: int(re.search(r"\." + f + r":\s+(\d+)", b).group(1))
    print(f"{dn[:90]:90s} vgpr {g('vgpr_count'):3d} sgpr {g('sgpr_count'):3d} vspill {g('vgpr_spill_count'):3d} sspill {g('sgpr_spill_count'):3d} lds {g('group_segment_fixed_size'):6d} scratch {g('private_segment_fixed_size')}")
