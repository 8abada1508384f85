"""Resource table (LDS / SGPR / VGPR / spills) of every kernel in an AMDGPU assembly listing:  python tools/kres.py file.s [regex]"""
import re, sys
txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
meta = txt[txt.index("amdhsa.kernels"):]
for blk in re.split(r"\n  - ", meta)[1:]:
    def g(k):
        m = re.search(r"\." + k + r":\s+(\S+)", blk)
        return m.group(1) if m else "?"
    name = g("name")
    if pat and not pat.search(name): continue
    print("%-70s lds %6s sgpr %3s sspill %3s vgpr %3s vspill %3s" % (name.replace("_ZN5srmap12_GLOBAL__N_1", "")[:70], g("group_segment_fixed_size"), g("sgpr_count"), g("sgpr_spill_count"), g("vgpr_count"), g("vgpr_spill_count")))
