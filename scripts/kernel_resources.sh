#!/bin/bash
# usage: bash scripts/kernel_resources.sh OBJECT.o [FILTER]  -- registers, spills, scratch and LDS of every kernel in a hipcc object
# (the code-object notes of its gfx950 device image)
O=$(realpath $1); F=${2:-.}
T=$(mktemp -d)
L=/opt/rocm/lib/llvm/bin
$L/llvm-objcopy --dump-section .hip_fatbin=$T/fb.bin $O
$L/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fb.bin --output=$T/dev.co --unbundle
$L/llvm-readelf --notes $T/dev.co | python3 -c '
import sys, re
txt = sys.stdin.read()
for blk in re.split(r"\n\s*- \.agpr_count", txt)[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    name = re.sub(r"^_ZN\d+_GLOBAL__N_\w+?\d+", "", g("name"))
    print("%-70s vgpr %3s sgpr %3s vspill %3s sspill %3s scratch %4s lds %6s" % (name[:70], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
' | grep -E "$F"
rm -rf $T
