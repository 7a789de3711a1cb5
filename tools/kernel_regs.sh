#!/bin/bash
# usage: tools/kernel_regs.sh [object]  -- VGPR/SGPR/spill/LDS metadata of every kernel in a code object
OBJ=${1:-sina_amd/csrc/build/mesh_dp.o}
TMP=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$TMP/fat.bin $OBJ
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$TMP/fat.bin --output=$TMP/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $TMP/dev.co | python3 -c '
import sys,re
name=None; d={}
for l in sys.stdin:
    l=l.strip()
    m=re.match(r"\.(name|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size|agpr_count):\s+(.*)",l)
    if m:
        d[m.group(1)]=m.group(2)
        if m.group(1)=="vgpr_spill_count":
            n=d.get("name","?")
            mm=re.search(r"mesh_dp_kernelILi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)",n)
            tag=("dp %sx%s rw%s w%s f%s"%mm.groups()) if mm else n[:60]
            print("%-28s vgpr %4s agpr %3s sgpr %4s  vspill %3s sspill %3s scratch %5s"%(tag,d.get("vgpr_count"),d.get("agpr_count","-"),d.get("sgpr_count"),d.get("vgpr_spill_count"),d.get("sgpr_spill_count"),d.get("private_segment_fixed_size")))
            d={}
'
rm -rf $TMP
