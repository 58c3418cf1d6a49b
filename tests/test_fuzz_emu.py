"""A short run of tools/fuzz_emu.py (randomised shapes through the streaming kernels on the CPU emulator) inside the CPU suite."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_streaming_kernels_on_random_shapes():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_emu.py"), "22", "3"], cwd=ROOT, capture_output=True,
                         text=True, timeout=1500)
    assert out.returncode == 0 and "fuzz_emu: 22 cases ok" in out.stdout, (out.stdout[-1500:], out.stderr[-1500:])
