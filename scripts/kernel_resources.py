"""Register / LDS / occupancy report of the kernels of one FFT size:
    python scripts/kernel_resources.py 12 [extra hipcc flags]      (inst_12.hip: workgroup-level kernels)
    python scripts/kernel_resources.py w12 [extra hipcc flags]     (instw_12.hip: wave-level kernels)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
l2 = sys.argv[1]
stem = "inst_"
if l2.startswith("w"):
    stem, l2 = "instw_", l2[1:]
l2 = f"{int(l2):02d}"
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-signed-zeros", "-fno-slp-vectorize",
       *sys.argv[2:], "-Rpass-analysis=kernel-resource-usage", "-c",
       os.path.join(ROOT, "lithographysimulator_amd", "csrc", f"{stem}{l2}.hip"), "-o", f"/tmp/inst_res_{l2}.o"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = cur.split("(")[0].replace("void litho::", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark: +([A-Za-z \[\]/]+): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
for k, d in rows.items():
    g = lambda key: d.get(key, "?")
    print(f"{k:46s} vgpr {g('VGPRs'):>4s} agpr {g('AGPRs'):>3s} sgpr {g('TotalSGPRs'):>4s} spill {g('VGPRs Spill'):>3s} "
          f"scratch {g('ScratchSize [bytes/lane]'):>4s} occ {g('Occupancy [waves/SIMD]'):>2s} lds {g('LDS Size [bytes/block]')}")
