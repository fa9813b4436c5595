"""Instruction census of one kernel of a HIP source between its workgroup barriers (no GPU needed):
    python tools/isa_census.py block_sliced.hip '_ZN12_GLOBAL__N_115block_fs_kernelILi2ELi4ELi4ELb0ELi1ELb0EEEvNS_6FsArgsE'
Compiles the source with the product flags to device assembly and counts, per barrier-separated segment of the named kernel, the
vector ALU instructions (with the packed, conversion and transcendental ones among them), MFMAs, LDS and vector-memory instructions,
waits and s_nops.  (Round 6: the accounting of the fused block kernel in DESIGN 4.1.)"""
import os
import re
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tante_amd.build import CSRC, EXTRA_FLAGS, FLAGS, HIPCC  # noqa: E402

src, name = sys.argv[1], sys.argv[2]
asm = "/tmp/_census.s"
subprocess.run([HIPCC, *FLAGS, *EXTRA_FLAGS.get(src, []), "--cuda-device-only", "-S", "-o", asm, os.path.join(CSRC, src)], check=True,
               capture_output=True)
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
end = start
while not lines[end].startswith(".Lfunc_end"):
    end += 1
segs = [Counter()]
trans = re.compile(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_")
for l in lines[start:end]:
    t = l.strip().split()
    if not t or t[0].startswith(";") or t[0].endswith(":") or t[0].startswith("."):
        continue
    op, c = t[0], segs[-1]
    if op == "s_barrier":
        segs.append(Counter())
        continue
    if op.startswith("v_mfma"):
        c["mfma"] += 1
    elif op.startswith("v_"):
        c["valu"] += 1
        c["valu_transcendental"] += bool(trans.match(op))
        c["valu_packed"] += op.startswith("v_pk_")
        c["valu_cvt"] += op.startswith("v_cvt")
    elif op.startswith("ds_"):
        c["lds"] += 1
    elif op.startswith(("global_", "buffer_", "scratch_")):
        c["vmem"] += 1
        c["scratch"] += op.startswith("scratch_")
    elif op.startswith("s_waitcnt"):
        c["waitcnt"] += 1
    elif op == "s_nop":
        c["s_nop"] += 1
    elif op.startswith("s_"):
        c["salu"] += 1
tot = Counter()
for i, c in enumerate(segs):
    tot.update(c)
    print(f"segment {i}: " + ", ".join(f"{k} {v}" for k, v in sorted(c.items()) if v))
print("total:     " + ", ".join(f"{k} {v}" for k, v in sorted(tot.items()) if v))
