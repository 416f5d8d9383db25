"""Post-pass over the device assembly hipcc emits (-S): every `v_cndmask_b32_e32 dst, a, b, vcc` whose sources the VOP3 encoding
can take (VGPRs and inline constants: gfx9's VOP3 has no literal and one constant-bus read, which the mask uses) becomes
`v_cndmask_b32_e64 dst, a, b, vcc`.  Same instruction, same result; on gfx950 a VOP2 select that directly follows another one
costs the SIMD ~17-22 cycles, the VOP3 form 4.4-6 (scripts/valu_throughput.hip, profiles/round6_valu_rate.txt), and hipcc emits
the VOP2 form whenever the mask sits in vcc — a 64-bit or float4 select is two to four of them in a row.
  python3 scripts/vop3_selects.py in.s out.s   -> prints how many it rewrote / left"""
import re
import sys

VGPR = r"v\d+"
OPERAND = r"(?:v\d+|-?\d+(?:\.\d+)?|0x[0-9a-fA-F]+)"
PAT = re.compile(rf"^(\s*)v_cndmask_b32_e32 ({VGPR}), ({OPERAND}), ({VGPR}), vcc\s*$")


def inline_ok(op):
    if op.startswith("v"):
        return True
    try:
        return -16 <= int(op, 0) <= 64
    except ValueError:
        return op.lstrip("-") in ("0.5", "1.0", "2.0", "4.0")


def main(src, dst):
    done = left = 0
    out = []
    for line in open(src):
        m = PAT.match(line.rstrip("\n"))
        if m and inline_ok(m.group(3)):
            out.append(f"{m.group(1)}v_cndmask_b32_e64 {m.group(2)}, {m.group(3)}, {m.group(4)}, vcc\n")
            done += 1
        else:
            if "v_cndmask_b32_e32" in line:
                left += 1
            out.append(line)
    open(dst, "w").writelines(out)
    print(f"{src}: {done} selects to VOP3, {left} left in VOP2")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
