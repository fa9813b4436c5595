"""Per-kernel register / scratch / LDS usage of one HIP source, compiled with the product flags (no GPU needed):
    python tools/kernel_resources.py block_sliced.hip [filter]
Reads hipcc -Rpass-analysis=kernel-resource-usage; a non-zero ScratchSize is a register spill."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tante_amd.build import CSRC, EXTRA_FLAGS, FLAGS, HIPCC  # noqa: E402

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = [HIPCC, *FLAGS, *EXTRA_FLAGS.get(src, []), "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", "/tmp/_kr.o"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r"  VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("sgpr", r"  SGPRs: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
for r in rows:
    if flt in r["name"]:
        print(f"{r['name']:<70s} vgpr {r.get('vgpr', -1):4d} agpr {r.get('agpr', 0):3d} sgpr {r.get('sgpr', -1):3d} scratch {r.get('scratch', -1):4d} "
              f"occ {r.get('occ', -1)} lds {r.get('lds', -1)}")
