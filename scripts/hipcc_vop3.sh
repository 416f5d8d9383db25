#!/bin/bash
# (experiment, not the product's build: +0.9 % on the headline, nothing on the frame builder — DESIGN.md §9)
# hipcc -c with a post-pass over the device assembly: VOP2 selects (v_cndmask_b32_e32 ..., vcc) re-encoded as VOP3
# (scripts/vop3_selects.py says why).  The same steps hipcc runs itself (hipcc -###), with the assembly in the middle:
#   device code -> .s -> post-pass -> assembler -> lld -> offload bundle -> host compile with the bundle embedded.
#   hipcc_vop3.sh SRC.hip OUT.o [hipcc flags ...]      (A3D_VOP3_KEEP=dir keeps the intermediate files there)
set -e
SRC=$1; OUT=$2; shift 2
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
LLVM=${LLVM:-/opt/rocm/lib/llvm/bin}
ARCH=${ARCH:-gfx950}
HERE="$(cd "$(dirname "$0")" && pwd)"
TMP=${A3D_VOP3_KEEP:-$(mktemp -d)}
mkdir -p "$TMP"
B="$TMP/$(basename "${OUT%.o}")"
$HIPCC "$@" --cuda-device-only -S "$SRC" -o "$B.s"
python3 "$HERE/vop3_selects.py" "$B.s" "$B.vop3.s" > "$B.log"
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=$ARCH -c "$B.vop3.s" -o "$B.dev.o"
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o "$B.hsaco" "$B.dev.o"
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--$ARCH \
  -input=/dev/null -input="$B.hsaco" -output="$B.hipfb"
$HIPCC "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$B.hipfb" -c "$SRC" -o "$OUT"
[ -n "$A3D_VOP3_KEEP" ] || rm -rf "$TMP"
