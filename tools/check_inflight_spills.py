"""ISA guard for the DP kernels' scalar loads.

History: until late in round 2 the row records / predecessor entries / edge records were fetched with
hand-placed `s_load_dwordx4` in inline asm.  The compiler takes an asm's outputs for ready when the asm
statement ends, so the B = 12 --insertion=forbid kernels -- shortest of SGPRs -- spilled the destination
registers right behind the load and restored garbage.  Since then every scalar load is the compiler's own
(a load from the constant address space, mesh_dp.hip sload16) and it tracks them in flight.

What this checks on the ISA of every DP kernel variant (hipcc -S --cuda-device-only; `make -C sina_amd/csrc isa`):
  1. NO scalar load is issued from inside an inline-asm block any more (the pattern that broke);
  2. no other memory instruction is either (s_load / global_ / ds_ / buffer_ in asm: same blindness);
  3. every kernel was actually looked at (a guard that sees nothing protects nothing): at least one
     compiler-issued s_load_dwordx4 per kernel, and the kernels named on the command line are present.
usage: tools/check_inflight_spills.py <file.s> [kernel-name-fragment ...]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
must_have = sys.argv[2:]
kern, inasm = None, False
res = {}  # kernel -> [asm memory instructions, compiler s_load_dwordx4]
for l in lines:
    if l.startswith("_ZN") and ":" in l and ("mesh_dp_kernel" in l or "mesh_dp_simple_kernel" in l):
        m = re.search(r"(mesh_dp(?:_simple)?_kernelI\w+?E)Ev", l)
        kern = m.group(1) if m else l.split(":")[0][:60]
        res.setdefault(kern, [0, 0])
        inasm = False
        continue
    if l.startswith("_ZN") and ":" in l:
        kern = None
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
    op = t.split()[0]
    is_mem = op.startswith(("s_load", "s_buffer_load", "global_", "ds_", "buffer_", "flat_", "scratch_"))
    if inasm and is_mem:
        res[kern][0] += 1
    if not inasm and op.startswith("s_load_dwordx4"):
        res[kern][1] += 1
bad = False
for k, (in_asm, own) in sorted(res.items()):
    flag = ""
    if in_asm:
        flag, bad = "  <-- memory instruction inside inline asm", True
    if own == 0:
        flag, bad = flag + "  <-- no scalar load seen: the guard looked at nothing", True
    print("%-44s asm-memory-ops %d  compiler-s_load_dwordx4 %d%s" % (k, in_asm, own, flag))
for want in must_have:
    if not any(want in k for k in res):
        print("kernel %s not found in %s" % (want, sys.argv[1]))
        bad = True
if not res:
    print("no DP kernel found")
    bad = True
sys.exit(1 if bad else 0)
