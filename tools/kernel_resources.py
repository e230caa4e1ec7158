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


_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def _vregs(text):
    regs = set()
    for m in _VREG.finditer(text):
        if m.group(1) is not None:
            regs.add(int(m.group(1)))
        else:
            regs.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return regs


def check_asm_loads(asm_path):
    """The hand-issued distance-field loads (inline asm `global_load_dwordx4`, csrc/gtop_kernels.hip asm_load_pair) are
    invisible to the compiler's wait-count insertion: the kernel waits for them itself (gtop_wait_pairs).  That is sound
    only while NOTHING touches a load's destination registers between its issue and the s_waitcnt that covers it — a
    v_mov the register allocator inserts after a compiler or flag change would read registers that have not landed.
    This walks the ISA of every kernel: each inline-asm load's destination VGPRs are tracked from its issue until an
    `s_waitcnt vmcnt(k)` leaves at most k memory operations outstanding (memory operations complete in order, so the
    oldest are done first; compiler-issued loads and stores in between are counted as outstanding too, which only
    makes the check stricter), and any instruction that names one of them in between is a violation.
    Returns (number of hand-issued loads seen, list of violations)."""
    kernel, in_asm, pending, seen, bad = None, False, [], 0, []   # pending: (dest VGPRs or None, line no, text)
    with open(asm_path) as f:
        for no, raw in enumerate(f, 1):
            line = raw.strip() if raw.lstrip().startswith(";;#") else raw.split(";")[0].strip()
            if line.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if line.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not line:
                continue
            if re.match(r"^[A-Za-z_.$][\w.$]*:", line):
                if not line.startswith(".L"):        # a new function: nothing is in flight across it
                    kernel, pending = line.split(":")[0], []
                continue
            if line.startswith("."):
                continue
            op = line.split()[0]
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", line)
                if m:
                    k = int(m.group(1))
                    pending = pending[len(pending) - k:] if 0 < k < len(pending) else ([] if k == 0 else pending)
                continue
            if op == "s_endpgm":
                pending = []
                continue
            touched = _vregs(line)
            for dest, pno, ptext in pending:
                if dest and dest & touched:
                    bad.append(f"{kernel}: line {no} `{line}` touches v{sorted(dest & touched)} of the load issued at "
                               f"line {pno} (`{ptext}`) before a wait covers it")
            if re.match(r"(global|flat|buffer|scratch)_(load|store|atomic)", op):
                if in_asm and op.startswith("global_load"):
                    seen += 1
                    pending.append((_vregs(line.split(",")[0]), no, line))
                else:
                    pending.append((None, no, line))
    return seen, bad


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
    seen, viol = check_asm_loads(asm_out)
    print(f"{seen} hand-issued loads in the ISA, {len(viol)} touched before their wait")
    for v in viol[:20]:
        print("  " + v)
    return 1 if bad or viol else 0


if __name__ == "__main__":
    sys.exit(main())
