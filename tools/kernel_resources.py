#!/usr/bin/env python3
"""Kernel count, registers, spills and occupancy of every kernel in csrc/gtop_kernels.hip (a CPU-side compile:
hipcc -Rpass-analysis=kernel-resource-usage); exits non-zero if any kernel spills — run by tests/test_capi.py, so a
compiler or flag change that pushes a body into scratch (the hand-issued loads of the latency variant depend on the
register allocator keeping their results where they land) is seen at build time.
usage: tools/kernel_resources.py [--asm-out FILE] [extra hipcc flags...]"""
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "grad_traj_optimization_amd", "csrc")


def analyse(extra=(), asm_out="/tmp/gtop_kernels.s"):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
           "-I" + CSRC, "-ffp-contract=on", "-mllvm", "-amdgpu-kernarg-preload-count=10", *extra, "-x", "hip",
           os.path.join(CSRC, "gtop_kernels.hip"), "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
           "-o", asm_out]
    t0 = time.time()
    out = subprocess.run(cmd, capture_output=True, text=True)
    secs = time.time() - t0
    if out.returncode != 0:
        raise RuntimeError(out.stderr[-2000:])
    rows = []
    for b in re.split(r"remark: Function Name: ", out.stderr)[1:]:
        name = b.split()[0]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]

        def g(k):
            return int(re.search(k + r": (\d+)", b).group(1))
        rows.append(dict(kernel=dem, vgprs=g("VGPRs"), sgprs=g("TotalSGPRs"), scratch=g(r"ScratchSize \[bytes/lane\]"),
                         sgpr_spill=g("SGPRs Spill"), vgpr_spill=g("VGPRs Spill"),
                         waves_per_simd=g(r"Occupancy \[waves/SIMD\]")))
    return rows, secs


def main():
    args = sys.argv[1:]
    asm_out = "/tmp/gtop_kernels.s"
    if args[:1] == ["--asm-out"]:
        asm_out, args = args[1], args[2:]
    rows, secs = analyse(args, asm_out)
    bad = 0
    for r in rows:
        spill = r["scratch"] or r["vgpr_spill"]      # to memory; SGPR "spills" go to VGPR lanes (v_writelane), listed only
        bad += bool(spill)
        print(f"{r['vgprs']:4d} VGPR {r['sgprs']:4d} SGPR  {r['waves_per_simd']} waves/SIMD  "
              f"scratch {r['scratch']:3d}  spilled SGPR/VGPR {r['sgpr_spill']:3d}/{r['vgpr_spill']:<3d} {r['kernel'][:130]}"
              + ("   <== SCRATCH" if spill else ""))
    print(f"{len(rows)} kernels, compiled in {secs:.1f} s, {bad} use scratch memory")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
