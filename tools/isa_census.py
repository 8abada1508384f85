"""Per-category instruction census of basic blocks of one kernel's ISA listing.
   python tools/isa_census.py <file.s> [--kernel SUBSTR] [--blocks L1,L2,...] [--weights w1,w2,...]
Without --blocks: one line per basic block.  With --blocks: the weighted sum over the named blocks (weights = execution
counts per wave, default 1) -- a dynamic census of a path through the kernel."""
import re, sys
def arg(n, d=None):
    return sys.argv[sys.argv.index(n) + 1] if n in sys.argv else d
path = sys.argv[1]
ksub = arg("--kernel")
CATS = ["f64", "cmpsel", "v32", "lane", "salu", "smem", "lds", "vmem", "wait"]
def cat(op):
    if op.startswith(("v_cmp", "v_cndmask")): return "cmpsel"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane"
    if op.startswith("v_"): return "f64" if "f64" in op or "b64" in op and op.startswith("v_mov") else "v32"
    if op == "s_waitcnt": return "wait"
    if op.startswith(("s_load", "s_buffer_load")): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    return None
blocks, order = {}, []
cur = "entry"; inside = ksub is None
blocks[cur] = dict.fromkeys(CATS, 0); order.append(cur)
for line in open(path):
    if ksub is not None:
        m = re.match(r"^(\S+):\s*(;.*)?$", line)
        if m and not m.group(1).startswith(".L"):
            inside = ksub in m.group(1)
            if inside: cur = "entry"; blocks = {cur: dict.fromkeys(CATS, 0)}; order = [cur]
            continue
        if line.startswith(".Lfunc_end") and inside: inside = False
    if not inside: continue
    m = re.match(r"^(\.LBB\w+):", line)
    if m:
        cur = m.group(1); blocks[cur] = dict.fromkeys(CATS, 0); order.append(cur); continue
    t = line.strip().split()
    if not t or t[0].startswith(";"): continue
    c = cat(t[0])
    if c: blocks[cur][c] += 1
sel = arg("--blocks")
if sel is None:
    print("%-12s " % "block" + " ".join("%6s" % c for c in CATS))
    for b in order:
        if sum(blocks[b].values()): print("%-12s " % b + " ".join("%6d" % blocks[b][c] for c in CATS))
else:
    names = sel.split(","); w = [float(x) for x in arg("--weights", ",".join(["1"] * len(names))).split(",")]
    tot = dict.fromkeys(CATS, 0.0)
    for n, ww in zip(names, w):
        b = blocks[n if n.startswith(".") or n == "entry" else ".LBB0_" + n]
        for c in CATS: tot[c] += ww * b[c]
    valu = tot["f64"] + tot["cmpsel"] + tot["v32"] + tot["lane"]
    print(" ".join("%s %.0f" % (c, tot[c]) for c in CATS), "| VALU %.0f" % valu)
