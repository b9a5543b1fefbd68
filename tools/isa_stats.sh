#!/bin/bash
# Per-kernel register / LDS / spill figures and VALU mix of one .hip file (gfx950 ISA, no GPU needed).
# usage: tools/isa_stats.sh csrc/kernels_primary_p2.hip [extra hipcc flags]
set -e
SRC=$1; shift
OUT=/tmp/isa_$(basename $SRC .hip).s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
    --cuda-device-only -S "$@" -o $OUT rust-wgpu-raytracing_amd/$SRC
python3 - $OUT <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(r"\.amdhsa_" + k + r"\s+(\S+)", body) or [None, "?"])[1]
    print(f"{name[:90]:90s} vgpr {g('next_free_vgpr'):>4s} sgpr {g('next_free_sgpr'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>5s} accum_off {g('accum_offset')}")
PY
echo "asm: $OUT"
