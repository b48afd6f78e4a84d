#!/bin/bash
# usage: profiles/tools/hotblock.sh [extra flags] -> per hot block stats of the default dev kernels
# Development aid (build machine, no GPU): compiles the bench instantiations of the fused carve kernel to assembly and prints,
# per run-loop block (the ones with >= 8 ds_read2_b32), instruction counts by kind -- in particular vector loads (gload,
# must be 0: the x-table records belong in scalar loads), s_waitcnt and SGPR spills through lanes.  Caught the -15 %
# regression of round 3 (a uniform value first computed in a divergent branch).
mkdir -p /tmp/vcy_asm; cd "$(dirname "$0")/../../vacancy_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -DVCY_DEV_BENCH_KERNELS_ONLY "$@" -I../../include -I. -S --cuda-device-only -o /tmp/vcy_asm/fused_dev.s ${SRC:-carve_fused_u8.hip} 2>&1 | grep -v hip-link
cd /tmp/vcy_asm
grep -E "^\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|name):" fused_dev.s | paste - - - - - - | grep -E "carve_fused" | sed 's/ \+/ /g;s/_ZN3vcy12_GLOBAL__N_1//;s/EEvNS.*//;s/.private_segment_fixed_size/priv/;s/.name: 18carve_fused_kernel//'
python3 - <<'P'
import re
txt=open("/tmp/vcy_asm/fused_dev.s").read()
for m in re.finditer(r"^(_ZN3vcy12_GLOBAL__N_118carve_fused_kernelI(\w+?)EEv\w*):", txt, re.M):
    start=m.end(); end=txt.index(".Lfunc_end", start)
    L=txt[start:end].split("\n")
    blocks=[];cur=[];name="entry"
    for l in L:
        mm=re.match(r"^(\.LBB\d+_\d+):",l)
        if mm:
            blocks.append((name,cur));name=mm.group(1);cur=[]
        elif l.startswith("\t") and not l.strip().startswith((";",".")): cur.append(l.strip())
    blocks.append((name,cur))
    print(m.group(2), "insts", sum(len(b) for _,b in blocks), "vmem-in-ds-blocks:", end=" ")
    for n,b in blocks:
        nd=sum(1 for i in b if i.startswith("ds_read2_b32"))
        if nd>=8:
            print("[%s n=%d valu=%d salu=%d wait=%d gload=%d sload=%d lanespill=%d]"%(n,len(b),sum(i.startswith("v_") for i in b),sum(i.startswith("s_") for i in b),sum(i.startswith("s_waitcnt") for i in b),sum(i.startswith("global_load") for i in b),sum(i.startswith("s_load") for i in b),sum(i.startswith(("v_writelane","v_readlane")) for i in b)), end=" ")
    print()
P
