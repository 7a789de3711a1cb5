"""Scans the DP kernels' ISA for the hazard of hand-placed asynchronous scalar loads: the destination
SGPRs of an `s_load` issued from inline asm being read (spilled with v_writelane, copied, used) before
an `s_waitcnt lgkmcnt(0)` -- the compiler believes an asm's outputs are ready when the asm ends.
usage: tools/check_inflight_spills.py <file.s>   (hipcc -S --cuda-device-only)"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
kern, inasm, pending, res = None, False, {}, {}
for l in lines:
    if l.startswith("_ZN") and ":" in l:
        m = re.search(r"mesh_dp_kernelI(\w+?)EEv", l)
        kern = m.group(1) if m else l.split(":")[0][:48]
        pending = {}
        res.setdefault(kern, 0)
        continue
    t = l.strip()
    if t.startswith(";;#ASMSTART"):
        inasm = True
        continue
    if t.startswith(";;#ASMEND"):
        inasm = False
        continue
    t = t.split(";")[0].strip()
    if not t or kern is None:
        continue
    if t.startswith(".LBB"):  # (a new block: keep what is pending -- fall-through is the common case)
        continue
    m = re.match(r"s_load_dwordx?\d* s\[(\d+):(\d+)\]", t)
    if m and inasm:
        pending[(int(m.group(1)), int(m.group(2)))] = True
        continue
    if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
        pending = {}
        continue
    ops = t.split(None, 1)
    if len(ops) < 2:
        continue
    parts = ops[1].split(",")
    srcs = ",".join(parts[1:]) if len(parts) > 1 else ""
    regs = set()
    for a, b in re.findall(r"s\[(\d+):(\d+)\]", srcs):
        regs.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bs(\d+)\b", srcs):
        regs.add(int(a))
    if any(a <= r <= b for (a, b) in pending for r in regs):
        res[kern] += 1
bad = {k: v for k, v in res.items() if v}
for k, v in res.items():
    print("%-28s %d" % (k, v))
sys.exit(1 if bad else 0)
