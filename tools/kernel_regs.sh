#!/bin/bash
# Register and spill counts of the pruned-walk kernels of one fp16-filter object (hipcc cross-compiles: no GPU needed), and the
# listing tools/vgpr_liveness.py reads.   usage: tools/kernel_regs.sh <KCAP: 4|8|12|16> [extra compiler flags]  -> build_ab/asm/f16_<KCAP>.s
k=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/build_ab/asm
out=$root/build_ab/asm/f16_$k.s
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-sched-strategy=max-ilp -DMCE_KCAP=$k -DMCE_INST_PART=1 "$@" -S --cuda-device-only -o $out $root/mcevidence_amd/csrc/knn_inst.hip 2>&1 | grep -v hip-link | tail -5
python3 - $out <<'P'
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'\.name:\s+(_ZN3mce14knn_f16_kernelILi1ELi\d+ELb1\S+)', s):
    blk = s[m.start():m.start() + 3000]
    g = lambda k: re.search(k + r':\s+(\d+)', blk).group(1)
    print(m.group(1)[:60], 'sgpr', g('.sgpr_count'), 'sgpr spills', g('.sgpr_spill_count'), 'vgpr', g('.vgpr_count'), 'vgpr spills', g('.vgpr_spill_count'))
P
