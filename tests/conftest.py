import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _cpu_workers(config):
    """Worker processes for a CPU-only run (pytest-xdist, no -n given): the emulator parity tests are ~10^3 x slower than the GPU's, and a plain
    `pytest tests -m "not gpu"` on the 8-core build container took 27 minutes serially.  Never on a box with a GPU (one device: the -m gpu tests run in
    one process), never inside a worker, FS_TEST_WORKERS=0 turns it off / =N pins it."""
    if hasattr(config, "workerinput") or not config.pluginmanager.hasplugin("xdist"):
        return 0
    if getattr(config.option, "numprocesses", None) is not None or config.getoption("collectonly", False) or config.getoption("usepdb", False):
        return 0
    env = os.environ.get("FS_TEST_WORKERS")
    if env is not None:
        return max(0, int(env))
    try:
        import torch
        if torch.cuda.is_available():
            return 0
    except Exception:
        pass
    return max(0, min(6, (os.cpu_count() or 1) - 2))   # (round 6: 4 -> 6 of the container's 8 cores; the sanitizer build and the two gloo ranks run beside the workers)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    n = _cpu_workers(config)
    if n > 1:
        # (xdist's own pytest_configure is trylast: it sees these values and starts the workers; the emulator library is built ONCE, here, before they exist)
        from tests import emu_lib
        emu_lib.build_emu()
        config.option.numprocesses = n
        config.option.dist = "load"
        config.option.tx = ["popen"] * n


# CPU runs: the tests that take minutes on the emulator go FIRST, longest first, so that the run ends when the work ends instead of when a ten-minute test
# that happened to be scheduled late does (measured durations of a `-m "not gpu"` run, seconds; a name matches by substring)
_SLOW_FIRST = [
    ("test_random_shapes_under_address_and_undefined_behaviour_sanitizers", 675), ("test_two_rank_sum_allreduce_equals_global_batch", 194),
    ("test_pool_gradient_routing_in_the_streaming_gram_gradient_kernel", 185), ("test_perceptual_loss_with_several_content_layers", 173),
    ("test_perceptual_loss_through_the_split_bf16_pipeline", 159), ("test_hip_train_step_matches_committed_fixture", 128),
    ("test_train_step_gradients_to_rounding_with_injected_masks", 116), ("test_tnet_narrow_layers_through_the_streaming_kernel", 113),
    ("test_tnet_residual_convs_through_the_half_item_winograd_kernel", 109), ("test_pool_gradient_routing_fused_into_the_gram_gradient_conv", 109),
    ("test_tnet_residual_convs_through_the_16tile_f4x4_kernel", 105), ("test_slow_style_steps_match_oracle", 105),
    ("test_streaming_kernels_on_random_shapes", 103), ("test_perceptual_loss_and_gradient_match_oracle", 86),
    ("test_tnet_residual_convs_through_the_winograd_kernel", 73), ("test_vgg_dgrad_named_export_matches_oracle", 66),
    ("test_tnet_sixteen_channel_layers_through_the_streaming_kernel", 65),
]


def pytest_collection_modifyitems(config, items):
    def weight(item):
        for name, w in _SLOW_FIRST:
            if name in item.nodeid:
                return -w
        return 0
    items.sort(key=weight)   # (stable: everything else keeps its order behind them)


@pytest.fixture(scope="session")
def repo_root():
    return ROOT
