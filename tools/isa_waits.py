"""Scan a gfx950 assembly listing (hipcc -S --cuda-device-only) for the pattern that cost the RoIAlign backward 20-30 % and the
deform_conv2d weight-gradient kernel its prefetch: a global load followed within a few instructions by `s_waitcnt vmcnt(0)` —
usually the copy that resolves a phi of a CONDITIONAL load (`x = 0; if (c) x = load`, `if (use_mask) m = load`), which makes the
load synchronous.  Prints, per kernel, how many loads are waited for at once, in loops and outside.
    python tools/isa_waits.py file.s [window=6]"""
import re, sys, subprocess
path = sys.argv[1]; win = int(sys.argv[2]) if len(sys.argv) > 2 else 6
name = None; rows = {}; recent = []; depth = 0
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name = m.group(1); rows[name] = [0, 0, 0]; recent = []; continue
    if name is None: continue
    t = line.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if "Loop Header" in t or "in Loop" in t: depth = 1
        elif t.startswith(".LBB") and "Loop" not in t: depth = 0
        continue
    op = t.split()[0]
    if op.startswith("global_load") or op.startswith("flat_load") or op.startswith("buffer_load"):
        rows[name][2] += 1; recent = [(win, depth)]
    elif op == "s_waitcnt" and "vmcnt(0)" in t and recent:
        rows[name][0 if recent[0][1] else 1] += 1; recent = []
    elif recent:
        w, d = recent[0]
        recent = [(w - 1, d)] if w > 1 else []
out = [(v[0], v[1], v[2], k) for k, v in rows.items() if v[0] + v[1]]
names = subprocess.run(["c++filt"], input="\n".join(o[3] for o in out), capture_output=True, text=True).stdout.split("\n")
print("in-loop  outside  loads  kernel")
for (a, b, c, _), n in sorted(zip(out, names), reverse=True):
    print(f"{a:7d} {b:8d} {c:6d}  {n[:130]}")
