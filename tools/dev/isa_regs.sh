#!/bin/bash
# Registers / scratch per kernel of one translation unit, from the ISA hipcc emits (nothing is left in csrc/):
#   tools/dev/isa_regs.sh beamform.hip [name filter]     (output under build_dev/isa/)
set -e
src=$1; pat=${2:-.}
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/build_dev/isa; mkdir -p $out
extra=""
[ "$src" = xylo.hip ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math --cuda-device-only -S $extra \
    -o $out/${src%.hip}.s $root/haghighatshoarmuir2024_amd/csrc/$src 2>/dev/null
python3 - "$out/${src%.hip}.s" "$pat" <<'PY'
import re, subprocess, sys
s = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2])
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name, body = m.group(1), m.group(2)
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r'\(.*', '', dn).replace('void micloc::', '')
    if not pat.search(dn):
        continue
    g = lambda k: (re.search(r'\.amdhsa_' + k + r'\s+(\S+)', body) or [None, '?'])[1]
    print(f"{dn:70s} vgpr {g('next_free_vgpr'):>4s} agpr_off {g('accum_offset'):>4s} scratch {g('private_segment_fixed_size'):>4s} lds {g('group_segment_fixed_size'):>6s}")
PY
