# Static instruction counts of one or more kernels of a source file as the Makefile's flags compile it (CPU only):
#   bash scripts/isa_kernel_stats.sh frame.hip level0_quad_kernelILb1E [more name fragments] [-- extra hipcc flags]
set -e
cd "$(dirname "$0")/.."
SRC=$1; shift
NAMES=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do NAMES+=("$1"); shift; done; [ "$1" = "--" ] && shift
OUT=${TMPDIR:-/tmp}/a3d_isa && mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -w --offload-arch=gfx950 "$@" \
  -Iinclude -S --cuda-device-only align3d_amd/csrc/$SRC -o $OUT/${SRC%.hip}.s
python3 - $OUT/${SRC%.hip}.s "${NAMES[@]}" <<'PY'
import re, sys, collections
s = open(sys.argv[1]).read().split('\n')
for name in sys.argv[2:]:
    starts = [i for i, l in enumerate(s) if re.match(r'^_Z\S*' + re.escape(name) + r'\S*:', l)]
    if not starts:
        print(name, 'not found'); continue
    start = starts[0]
    end = [i for i in range(start, len(s)) if s[i].startswith('.Lfunc_end')][0]
    ops = [l.split()[0] for l in s[start:end] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = collections.Counter(ops)
    tot = lambda pred: sum(v for k, v in c.items() if pred(k))
    meta = ' '.join(l.strip('; \t') for l in s[end:end + 80] if any(x in l for x in ('; NumVgprs', '; Occupancy', '; ScratchSize', '; LDSByteSize')))
    print(f"{name}: total {len(ops)} VALU {tot(lambda k: k.startswith('v_'))} (f64 {tot(lambda k: k.startswith('v_') and 'f64' in k)}) "
          f"SALU {tot(lambda k: k.startswith('s_'))} LDS {tot(lambda k: k.startswith('ds_'))} VMEM {tot(lambda k: k.startswith('global_') or k.startswith('buffer_'))} | {meta}")
    open(sys.argv[1].replace('.s', '.' + re.sub(r'\W', '_', name) + '.s'), 'w').write('\n'.join(s[start:end]))
PY
