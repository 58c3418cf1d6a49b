"""Engine fixtures: the same parity test bodies run against
  * "emu": the kernel sources compiled for the CPU fiber emulator (tests/emu) -- CPU suite;
  * "hip": the product libfaststyle_hip.so on a real MI355X -- `-m gpu` suite.
"""
import pytest


def _hip_engine():
    from faststyle_amd import engine
    return engine.Engine()          # raises loudly if the .so or the GPU is missing


def engine_params():
    return [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


_cache = {}


def get_engine(kind):
    if kind not in _cache:
        if kind == "emu":
            from tests import emu_lib
            _cache[kind] = emu_lib.emu_engine()
        else:
            _cache[kind] = _hip_engine()
    return _cache[kind]


def on_emulator(eng):
    """The CPU fiber emulator (tests/emu): ~10^3 x slower than the GPU, so a few tests take their SMALLER shape there (same code paths)."""
    return type(eng.mem).__name__ == "NumpyMem"
