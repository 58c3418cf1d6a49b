"""A short run of tools/fuzz_emu.py (randomised shapes through the streaming kernels on the CPU emulator) inside the CPU suite --
once against the normal emulator build, once against the same sources built with -fsanitize=address,undefined (SURVEY.md
section 5's sanitizer build: an out-of-range LDS / global index in a ragged-edge path is then an error even where it does not
change a compared value)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_streaming_kernels_on_random_shapes():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_emu.py"), "20", "3"], cwd=ROOT, capture_output=True,
                         text=True, timeout=1500)
    assert out.returncode == 0 and "fuzz_emu: 20 cases ok" in out.stdout, (out.stdout[-1500:], out.stderr[-1500:])


def test_random_shapes_under_address_and_undefined_behaviour_sanitizers():
    from tests import emu_lib
    emu_lib.build_emu_sanitized()
    env = dict(os.environ, FS_EMU_SANITIZE="1", LD_PRELOAD=emu_lib.ASAN_RT,
               ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1:verify_asan_link_order=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_emu.py"), "10", "7"], cwd=ROOT, capture_output=True,
                         text=True, timeout=2400, env=env)
    assert out.returncode == 0 and "fuzz_emu: 10 cases ok" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error:" not in out.stderr, out.stderr[-3000:]
