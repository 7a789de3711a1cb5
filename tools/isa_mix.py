"""Opcode mix of a DP kernel variant's ROW LOOP (the loop over DAG rows inside the loop over strips).

usage: tools/isa_mix.py <file.s> [variant ...]      (hipcc -S --cuda-device-only; `make -C sina_amd/csrc isa`)
       variant = template arguments as B,W,F,BELOW,DBG  e.g. 8,0,0,1,0, or simple,B,DBG for
       mesh_dp_simple_kernel (default: every non-DBG variant)

The row loop is found structurally: among the loops of the kernel (a backward branch to a label),
the largest one that is nested inside another loop (the strip loop).  Static counts over ALL paths of
the loop body (rare ones included: spill-row reads, the log-step scan, the end-cell search), so the
numbers are an upper bound of what one row executes; they are meant for comparing builds, not as a
cycle model.  Classes:
  valu      v_* except moves / readlane / writelane
  v_mov     v_mov_b32 / v_accvgpr_* split by source: vgpr, sgpr, const
  lane      v_readlane / v_writelane / v_readfirstlane (SGPR spill traffic and uniform broadcasts)
  dpp       VALU instructions with a DPP modifier (counted in valu too)
  salu      s_* except nop / waitcnt / branch
  s_nop, s_waitcnt, branch, lds (ds_*), smem (s_load*), vmem (global_* / scratch_* / buffer_*)
"""
import re
import sys
from collections import Counter


def kernels(lines):
    """[(variant tuple, first line, last line)] of every mesh_dp_kernel in the file"""
    out = []
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_ZN") and "mesh_dp_kernel" in l and ":" in l:
            m = re.search(r"mesh_dp_kernelILi(\d+)ELb(\d)ELb(\d)ELb(\d)ELb(\d)E", l)
            start = (tuple(int(x) for x in m.groups()), i)
        elif l.startswith("_ZN") and "mesh_dp_simple_kernel" in l and ":" in l:
            m = re.search(r"mesh_dp_simple_kernelILi(\d+)ELb(\d)ELb(\d)E", l)
            start = (("simple", int(m.group(1)), int(m.group(2)), int(m.group(3))), i)
        elif start and l.strip().startswith("s_endpgm"):
            out.append((start[0], start[1], i))
            start = None
    return out


def instrs(lines, a, b):
    """[(text, label or None)] -- instructions and labels in order"""
    out = []
    for l in lines[a:b + 1]:
        t = l.split(";")[0].strip()
        if not t or t.startswith(";"):
            continue
        if t.endswith(":"):
            if t.startswith(".LBB"):
                out.append((None, t[:-1]))
            continue
        if t.startswith("."):
            continue
        out.append((t, None))
    return out


def find_row_loop(ins):
    label_at = {lab: i for i, (t, lab) in enumerate(ins) if lab}
    loops = []
    for i, (t, lab) in enumerate(ins):
        if t and (t.startswith("s_cbranch") or t.startswith("s_branch")):
            tgt = t.split()[-1]
            if tgt in label_at and label_at[tgt] < i:
                loops.append((label_at[tgt], i))
    nested = [(a, b) for (a, b) in loops if any(c <= a and b <= d and (c, d) != (a, b) for (c, d) in loops)]
    if not nested:
        nested = loops
    return max(nested, key=lambda ab: ab[1] - ab[0])


def classify(t, c):
    op = t.split()[0]
    if op.startswith("v_"):
        if op in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"):
            c["lane"] += 1
            if op != "v_readfirstlane_b32":
                c["lane_spill"] += 1
            return
        c["valu_all"] += 1
        if op.startswith("v_mov_b32") or op.startswith("v_accvgpr") or op.startswith("v_mov_b64"):
            src = t.split(",")[-1].strip().split()[0]
            if "dpp" in t or "row_" in t or "wave_" in t:
                c["dpp"] += 1
                c["v_mov dpp"] += 1
            elif src.startswith("v") or src.startswith("a"):
                c["v_mov vgpr"] += 1
            elif src.startswith("s") or src in ("vcc_lo", "vcc_hi", "exec_lo", "exec_hi", "m0"):
                c["v_mov sgpr"] += 1
            else:
                c["v_mov const"] += 1
            return
        if "dpp" in t or "row_" in t or "wave_" in t:
            c["dpp"] += 1
        c["valu"] += 1
        c["op " + op.replace("_e32", "").replace("_e64", "")] += 1
    elif op == "s_nop":
        c["s_nop"] += 1
    elif op == "s_waitcnt":
        c["s_waitcnt"] += 1
    elif op.startswith("s_cbranch") or op.startswith("s_branch"):
        c["branch"] += 1
    elif op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_dcache"):
        c["smem"] += 1
    elif op.startswith("s_"):
        c["salu"] += 1
    elif op.startswith("ds_"):
        c["lds"] += 1
    elif op.startswith(("global_", "scratch_", "buffer_", "flat_")):
        c["vmem"] += 1
    else:
        c["other"] += 1


def main():
    lines = open(sys.argv[1]).read().split("\n")
    want = [tuple(x if x == "simple" else int(x) for x in a.split(",")) for a in sys.argv[2:]]
    for var, a, b in kernels(lines):
        if want and var not in want:
            continue
        if not want and (var[2] if var[0] == "simple" else var[-1]):  # (debug-plane variants)
            continue
        ins = instrs(lines, a, b)
        la, lb = find_row_loop(ins)
        c = Counter()
        n = 0
        for t, lab in ins[la:lb + 1]:
            if t:
                n += 1
                classify(t, c)
        mov = c["v_mov vgpr"] + c["v_mov sgpr"] + c["v_mov const"]
        meta = {}
        for l in lines[b:b + 80]:
            mm = re.match(r"; (NumVgprs|NumSgprs|SGPRSpill|ScratchSize|Occupancy|sgpr_spill_count): (\d+)", l.strip())
            if mm:
                meta.setdefault(mm.group(1), mm.group(2))
        name = "mesh_dp_simple_kernel<%d,%d,%d>" % var[1:] if var[0] == "simple" else "mesh_dp_kernel<%d,%d,%d,%d,%d>" % var
        print("%s row loop: %d instructions   [VGPRs %s, waves/SIMD %s, scratch %s]" % (
            name, n, meta.get("NumVgprs"), meta.get("Occupancy"), meta.get("ScratchSize")))
        print("  VALU %d (arith %d, v_mov %d = vgpr %d + sgpr %d + const %d, dpp moves %d)  lane ops %d (spill traffic %d)" % (
            c["valu_all"], c["valu"], mov, c["v_mov vgpr"], c["v_mov sgpr"], c["v_mov const"], c["v_mov dpp"], c["lane"],
            c["lane_spill"]))
        print("  SALU %d  s_nop %d  s_waitcnt %d  branch %d  LDS %d  SMEM %d  VMEM %d" % (
            c["salu"], c["s_nop"], c["s_waitcnt"], c["branch"], c["lds"], c["smem"], c["vmem"]))
        ops = sorted(((v, k[3:]) for k, v in c.items() if k.startswith("op ")), reverse=True)
        print("  top VALU opcodes: " + "  ".join("%s %d" % (k, v) for v, k in ops[:14]))


if __name__ == "__main__":
    main()
