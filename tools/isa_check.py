#!/usr/bin/env python
"""Address-space check of the device code: compiles every translation unit of dm-vio_amd/csrc to gfx950 ISA (hipcc -S --cuda-device-only, the Makefile's flags) and counts, per
kernel, the memory instructions by address space.  A pointer that does not arrive as a kernel argument is a GENERIC pointer to the compiler: accesses through it become flat_load /
flat_store, which count on BOTH wait counters (an s_waitcnt lgkmcnt(0) in front of an LDS read then also waits for every outstanding global load) — round 6 found the tracker's
image taps and every batched BA kernel in that state.  usage: python tools/isa_check.py [--json]; tests/test_isa_cpu.py asserts on the result."""
import json, os, re, subprocess, sys, tempfile
from collections import Counter
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dm-vio_amd", "csrc")
UNITS = ["capi", "capi_ba", "capi_immature", "capi_init"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]
KINDS = ("global_load", "global_store", "global_atomic", "flat_load", "flat_store", "flat_atomic", "scratch_load", "scratch_store")


def unit_isa(unit, outdir):
    out = os.path.join(outdir, unit + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-S", "--cuda-device-only", "-o", out, unit + ".hip"], cwd=CSRC, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read().splitlines()


def kernels(lines):
    """{demangled-ish kernel name: Counter of memory instruction kinds}"""
    res = {}
    starts = [(i, l) for i, l in enumerate(lines) if re.match(r"^_ZN3dmv\d+k_\w+:", l)]
    for i, l in starts:
        sym = l.split(":")[0]
        m = re.match(r"^_ZN3dmv(\d+)(k_\w+)", sym)
        name = m.group(2)[:int(m.group(1))] + sym[len("_ZN3dmv") + len(m.group(1)) + int(m.group(1)):][:24]
        c = Counter()
        for x in lines[i:]:
            if x.startswith(".Lfunc_end"):
                break
            mm = re.match(r"\s+(global_load|global_store|global_atomic|flat_load|flat_store|flat_atomic|scratch_load|scratch_store)", x)
            if mm:
                c[mm.group(1)] += 1
        res[name] = c
    return res


def run():
    with tempfile.TemporaryDirectory() as d:
        with ThreadPoolExecutor(len(UNITS)) as ex:
            isa = list(ex.map(lambda u: unit_isa(u, d), UNITS))
    out = {}
    for u, lines in zip(UNITS, isa):
        for k, c in kernels(lines).items():
            out[u + ":" + k] = {kk: c.get(kk, 0) for kk in KINDS}
    return out


if __name__ == "__main__":
    r = run()
    if "--json" in sys.argv:
        print(json.dumps(r))
    else:
        print("| kernel | global ld / st / atomic | flat ld / st / atomic | scratch ld / st |\n|---|---|---|---|")
        for k, c in sorted(r.items()):
            print("| %s | %d / %d / %d | %d / %d / %d | %d / %d |" % (k, c["global_load"], c["global_store"], c["global_atomic"], c["flat_load"], c["flat_store"], c["flat_atomic"],
                                                                     c["scratch_load"], c["scratch_store"]))
