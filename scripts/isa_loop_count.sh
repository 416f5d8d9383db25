# Static instruction counts of the ImageIcp pixel loop (image_icp_head_kernel<true> by default, two pixels per trip) as the
# Makefile's flags compile it: VALU / SALU / memory instructions, s_nop, VGPRs.  CPU only (hipcc -S).
#   bash scripts/isa_loop_count.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")/.."
OUT=${TMPDIR:-/tmp}/a3d_isa && mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize --offload-arch=gfx950 "$@" \
  -Iinclude -S --cuda-device-only align3d_amd/csrc/image_icp.hip -o $OUT/image_icp.s 2>/dev/null
KERNEL=${KERNEL:-_ZN12_GLOBAL__N_121image_icp_head_kernelILb1EE}  # (KERNEL=_ZN12_GLOBAL__N_116image_icp_kernelILi1ELb0E: the last-block form)
n=$(grep -n "^${KERNEL}.*:" $OUT/image_icp.s | head -1 | cut -d: -f1)
awk -v n=$n 'NR>=n' $OUT/image_icp.s | awk '/^\.Lfunc_end/{exit} {print}' > $OUT/kernel.s
st=$(grep -n "Loop Header: Depth=1" $OUT/kernel.s | head -1 | cut -d: -f1)
en=$(awk -v s=$st 'NR>=s' $OUT/kernel.s | grep -n "s_cbranch_vccz\|s_cbranch_scc0" | head -1 | cut -d: -f1)
sed -n "${st},$((st+en))p" $OUT/kernel.s | grep -v '^\s*;' | grep -v '^\.' | awk '{print $1}' > $OUT/loop_ops.txt
echo "per trip (2 pixels): VALU $(grep -c '^v_' $OUT/loop_ops.txt)  SALU $(grep -c '^s_' $OUT/loop_ops.txt) (s_nop $(grep -c '^s_nop' $OUT/loop_ops.txt), s_waitcnt $(grep -c '^s_waitcnt' $OUT/loop_ops.txt))  memory $(grep -c '^global_\|^ds_' $OUT/loop_ops.txt)  |  $(awk -v n=$n 'NR>=n' $OUT/image_icp.s | grep -m1 '; NumVgprs')  $(awk -v n=$n 'NR>=n' $OUT/image_icp.s | grep -m1 '; Occupancy')"
