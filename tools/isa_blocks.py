"""Per-basic-block instruction counts of one kernel's ISA listing:  python tools/isa_blocks.py <file.s>"""
import re, sys
name = "entry"; cnt = {}; order = []
def flush():
    pass
blocks = []
cur = {"name": "entry", "v": 0, "v64": 0, "s": 0, "ds": 0, "g": 0, "w": 0, "br": []}
for line in open(sys.argv[1]):
    m = re.match(r"^(\.LBB\w+):", line)
    if m:
        blocks.append(cur); cur = {"name": m.group(1), "v": 0, "v64": 0, "s": 0, "ds": 0, "g": 0, "w": 0, "br": []}
        continue
    t = line.strip().split()
    if not t: continue
    op = t[0]
    if op.startswith("v_"):
        cur["v"] += 1
        if "f64" in op: cur["v64"] += 1
    elif op.startswith("s_"):
        cur["s"] += 1
        if op == "s_waitcnt": cur["w"] += 1
        if op.startswith("s_cbranch") or op in ("s_branch", "s_barrier", "s_endpgm"): cur["br"].append(op.replace("s_cbranch_", "") + (":" + t[1] if len(t) > 1 else ""))
    elif op.startswith("ds_"): cur["ds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): cur["g"] += 1
blocks.append(cur)
for b in blocks:
    print("%-12s v=%4d (f64 %4d) s=%4d ds=%3d g=%3d wait=%2d  %s" % (b["name"], b["v"], b["v64"], b["s"], b["ds"], b["g"], b["w"], " ".join(b["br"])))
