"""The rendezvous layer's host logic (csrc/gtop_rendezvous.cpp: N serial NLopt-style callers sharing one launch — the
reference runs one optimizer per problem, src/grad_traj_optimizer.cpp:137-195) under ThreadSanitizer on the CPU: the
file is pure host code over one C-ABI entry, so it is linked here with a test double of that entry
(tests/cpp/rendezvous_tsan.cpp) and driven through every protocol path from real threads.  The GPU tests of the same
layer are in tests/test_rendezvous.py."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rendezvous_protocol_is_race_free_under_thread_sanitizer(tmp_path):
    exe = str(tmp_path / "rendezvous_tsan")
    csrc = os.path.join(ROOT, "grad_traj_optimization_amd", "csrc")
    subprocess.check_call(["g++", "-fsanitize=thread", "-O1", "-g", "-std=c++17", "-pthread",
                           "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                           os.path.join(ROOT, "tests", "cpp", "rendezvous_tsan.cpp"),
                           os.path.join(csrc, "gtop_rendezvous.cpp"), "-o", exe])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    for _ in range(3):          # (thread interleavings differ from run to run)
        out = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
        assert out.returncode == 0 and "rendezvous_tsan: ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
        assert "ThreadSanitizer" not in out.stderr, out.stderr[-4000:]
