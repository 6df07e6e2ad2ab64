#!/usr/bin/env python
"""Per-kernel register / scratch / LDS figures of the gfx950 code objects (hipcc -Rpass-analysis=kernel-resource-usage), as a table.
   usage: python tools/resource_usage.py [capi.hip ...] [--filter substr]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(ROOT, "dm-vio_amd", "csrc")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
flt = None
if "--filter" in sys.argv:
    flt = sys.argv[sys.argv.index("--filter") + 1]; args = [a for a in args if a != flt]
srcs = args or ["capi.hip", "capi_ba.hip", "capi_immature.hip", "capi_init.hip"]
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp".split()
print("| kernel | SGPRs | VGPRs | AGPRs | spilled SGPRs | spilled VGPRs | scratch B/lane | waves/SIMD | LDS B |"); print("|---|---|---|---|---|---|---|---|---|")
for s in srcs:
    p = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", s], cwd=C, capture_output=True, text=True)
    cur = None
    for l in p.stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", l)
        if not m: continue
        k, v = m.groups()
        if k in ("Function Name", "Name"):
            cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]}
        elif cur is not None:
            cur[k] = v
            if k.startswith("LDS"):
                if flt is None or flt in cur["name"]:
                    print("| `%s` | %s | %s | %s | %s | %s | %s | %s | %s |" % (cur["name"].replace("void dmv::", ""), cur.get("TotalSGPRs"), cur.get("VGPRs"), cur.get("AGPRs"), cur.get("SGPRs Spill"), cur.get("VGPRs Spill"),
                                                                      cur.get("ScratchSize [bytes/lane]"), cur.get("Occupancy [waves/SIMD]"), v))
                cur = None
