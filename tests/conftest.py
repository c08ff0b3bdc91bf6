import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# CPU tests of 10 s and more here (full-width oracle runs, world-size-2 process groups, whole runner jobs): `-m "not gpu and not slow"` is
# the two-minute tier for the edit-test loop; the driver's `-m "not gpu"` runs everything.
SLOW = {
    "test_full_width_fixture_pins_the_oracle_hooks_at_config1", "test_fused_runner_world2_deals_whole_clips_and_pipelines_them",
    "test_step_engines_are_reused_across_clips_without_stale_state", "test_frame_parallel_unet_world2_gloo_cpu",
    "test_pipelined_fused_runner_writes_what_the_serial_one_writes", "test_pipeline_loops_vs_oracle",
    "test_frame_parallel_runners_world2_match_single_process", "test_sharded_runners_world2_gather_every_entry",
    "test_source_feature_cache_multi_edit_is_bit_equal", "test_no_spill_or_copy_of_a_pending_asm_lds_read",
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a CPU test of 10 s or more (deselect with -m 'not gpu and not slow')")


def pytest_collection_modifyitems(config, items):
    import torch
    for item in items:
        if item.originalname in SLOW or item.name in SLOW:
            item.add_marker(pytest.mark.slow)
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
