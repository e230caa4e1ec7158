"""CPU-side checks of the drop-in boundary: libgtop_hip.so loads and exports
every symbol include/gtop.h declares; without a GPU it refuses to work instead
of falling back to a CPU path."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "gtop.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gtop_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_boundary():
    names = declared_functions()
    for must in ("gtop_create", "gtop_destroy", "gtop_set_params", "gtop_set_sdf", "gtop_set_problem",
                 "gtop_eval_batch", "gtop_cost_nlopt", "gtop_eval_device", "gtop_init_sdf_map",
                 "gtop_update_sdf_map"):
        assert must in names


def test_library_exports_every_declared_symbol(gtop):
    lib = ctypes.CDLL(gtop.library_path())
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} is declared in include/gtop.h but not exported"
    assert lib.gtop_abi_version() >= 1


def test_binding_covers_every_declared_symbol(gtop):
    lib = gtop.load_library()
    for name in declared_functions():
        f = getattr(lib, name)
        assert f.argtypes is not None, f"{name} has no ctypes signature in _lib.py"


def test_no_cpu_fallback(gtop):
    """On a box without a GPU the product path must fail loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(gtop.GtopError) as ei:
        gtop.GtopContext(device=0)
    assert ei.value.code == 3   # GTOP_ERR_NO_DEVICE


def test_product_does_not_touch_the_oracle():
    """oracle/ is test infrastructure: nothing under the product package may
    import, link or load it."""
    pkg = os.path.join(ROOT, "grad_traj_optimization_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "gtop_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f
    out = os.popen(f"ldd {os.path.join(pkg, 'libgtop_hip.so')}").read()
    assert "oracle" not in out


def test_no_kernel_spills_to_scratch_memory(tmp_path):
    """Build-time check (hipcc -Rpass-analysis=kernel-resource-usage, tools/kernel_resources.py): no instantiation of
    the evaluation kernel uses scratch memory.  The latency variant issues its distance-field loads through inline asm
    whose outstanding results the compiler cannot see; that is only sound while the register allocator keeps them
    where they land, i.e. while nothing spills — a compiler or flag change that breaks it fails here, not silently."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    asm = str(tmp_path / "gtop_kernels.s")       # a 5 MB ISA dump: scratch, never in the source tree
    rows, _ = kr.analyse(asm_out=asm)
    assert len(rows) >= 60
    bad = [r for r in rows if r["scratch"] or r["vgpr_spill"]]
    assert not bad, bad
    hot = [r for r in rows if r["kernel"].startswith("gtop_eval_wave_kernel<double, false, 3, 1, true, 2, GtopNoMma, false, false, 1>")]
    assert len(hot) == 1 and hot[0]["sgpr_spill"] == 0 and hot[0]["waves_per_simd"] == 2, hot
    # the same dump, instruction by instruction: nothing names a hand-issued load's destination registers between its
    # issue and the s_waitcnt that covers it (spill-free is necessary for that, not sufficient: a v_mov would do)
    seen, violations = kr.check_asm_loads(asm)
    assert seen >= 12 and not violations, violations[:5]
    # and the checker itself sees what it is there to see
    fake = tmp_path / "fake.s"
    fake.write_text("k:\n\t;;#ASMSTART\n\tglobal_load_dwordx4 v[4:7], v1, s[2:3]\n\t;;#ASMEND\n"
                    "\t;;#ASMSTART\n\tglobal_load_dwordx4 v[8:11], v1, s[2:3]\n\t;;#ASMEND\n"
                    "\ts_waitcnt vmcnt(1)\n\tv_add_f64 v[20:21], v[4:5], v[6:7]\n\tv_mov_b32_e32 v30, v9\n"
                    "\ts_waitcnt vmcnt(0)\n\tv_mov_b32_e32 v31, v10\n\ts_endpgm\n")
    seen, violations = kr.check_asm_loads(str(fake))
    assert seen == 2 and len(violations) == 1 and "v_mov_b32_e32 v30, v9" in violations[0], violations


def test_eigen_adapter_header_says_what_it_needs(tmp_path):
    """include/grad_traj_optimization/grad_traj_optimizer.h (the reference-shaped class with Eigen signatures) must fail
    with a readable message in an image without Eigen, and compile against the test double of the Eigen types
    (tests/cpp/eigen_double) — build() makes gtop_eigen_adapter from it."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    src = tmp_path / "t.cpp"
    src.write_text('#include "grad_traj_optimization/grad_traj_optimizer.h"\nint main() { return 0; }\n')
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "grad_traj_optimization_amd", "csrc")]
    bare = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", *inc, str(src)], capture_output=True, text=True)
    if bare.returncode == 0:
        pytest.skip("this image has Eigen")
    assert "needs Eigen3" in bare.stderr
    ok = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I" + os.path.join(ROOT, "tests", "cpp", "eigen_double"),
                         *inc, str(src)], capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr
    assert os.path.exists(os.path.join(ROOT, "grad_traj_optimization_amd", "gtop_eigen_adapter"))


def test_reciprocal_divisions_of_the_per_wavefront_task_lists_are_exact():
    """csrc/gtop_kernels.hip, the geometries with 21 / m or 64 / m trajectories per wavefront: a lane's trajectory is
    slot / m by a 16-bit reciprocal and a task's trajectory qi / n by a 24-bit one (no integer division in the kernel).
    Exhaustively, for every segment count and every index the kernel can form: the same as floor division, and the
    product stays inside 32 bits."""
    for slots in (64, 21):
        for m in range(2, slots + 1):
            inv = (65536 + m - 1) // m
            assert all(((slot * inv) >> 16) == slot // m for slot in range(64))
            n, nt = 9 * (m - 1), slots // m
            inv_n = ((1 << 24) + n - 1) // n
            for qi in range(nt * n):
                assert qi * inv_n < 2 ** 32 and ((qi * inv_n) >> 24) == qi // n
