"""`-m gpu` parity tests: HIP kernels / native UNet / pipeline loops vs PyTorch fp32, the CPU oracle and the golden
fixtures generated from the reference's own code.  All calls go through the C ABI (anyv2v_amd.ops -> ctypes)."""
import os

import pytest

import gpu_checks as gc

pytestmark = pytest.mark.gpu


def _assert_all(results):
    if os.environ.get("ANYV2V_PRINT_ROWS", "0") == "1":   # every row's measured error in the log (-s), not only the failures
        for r in results:
            print(f"{'ok  ' if r['ok'] else 'FAIL'} {r['name']}: {r['err']:.3e} (l2 {r.get('l2', float('nan')):.3e}, tol {r['tol']:.1e})")
    bad = [f"{r['name']}: err {r['err']:.3e} (l2 {r.get('l2', float('nan')):.3e}) > tol {r['tol']:.1e}" for r in results
           if not r["ok"] and not r.get("informational")]
    assert not bad, "\n".join(bad)


def test_hip_library_loaded_and_mfma_layouts():
    from anyv2v_amd import _lib
    assert _lib.load().anyv2v_version() >= 103
    _assert_all(gc.check_selftest())


@pytest.mark.parametrize("variant", ["reg", "glds", "naive"])
def test_gemm(variant):
    _assert_all(gc.check_gemm((variant,)))


def test_gemm_persistent_big_tile():
    _assert_all(gc.check_gemm_big())


@pytest.mark.parametrize("rows", [192, 256])
def test_gemm_pingpong_kernel(rows):
    """gemm_pp_kernel (round 4: the two waves of a SIMD run the K-tile's read / matrix slots one barrier apart), forced onto every
    non-GEGLU case of the persistent-kernel check with 192- and 256-row tiles: torch fp32 references, the naive kernel, and BIT-equality
    with the 128-row kernel (same MFMA shape, same K order), incl. launches with more tiles than CUs and several K-tiles per tile."""
    _assert_all(gc.check_gemm_big((1 << 17) | ((1 << 19) if rows == 192 else (1 << 20)), tag=f"pp{rows}"))


def test_gemm_single_wave_per_simd_kernel():
    """gemm_sw_kernel (round 6: four waves = one per SIMD, 96 x 160 wave tiles with the accumulators in AGPRs, barrier inside the MFMA
    stream, outputs stored straight from the registers through a permuted W row order), forced onto every eligible case of the
    persistent-kernel check: torch fp32 references, the naive kernel, BIT-equality with the 128-row kernel (same MFMA shape and K order),
    every tile order of the rastered launches, launches with more tiles than CUs, ragged M, GEGLU, two-source K loops."""
    _assert_all(gc.check_gemm_big(1 << 21, tag="sw"))


def test_conv3x3_lds_patch_reuse_kernel():
    _assert_all(gc.check_conv_halo())


def test_gemm_single_wave_kernel_stream_k():
    """The stream-K form of gemm_sw_kernel (blocks own contiguous ranges of (tile, K-tile) units; tiles cut by a range boundary are
    summed from fp32 slabs by gemm_sw_fixup_kernel), forced (flags bit27) onto every eligible case of the persistent-kernel check:
    torch fp32 references and the naive kernel at the usual tolerance; against the unsplit kernels 5e-4 (a different fp32 summation
    order, not bit-equal by construction)."""
    _assert_all(gc.check_gemm_big(1 << 27, tag="sk", exact=False))


def test_fused_feed_forward_c320():
    _assert_all(gc.check_ff_fused())


def test_gemm_weight_stationary_k320():
    _assert_all(gc.check_gemm_ws())


def test_gemm_weight_stationary_with_layernorm_fold():
    _assert_all(gc.check_gemm_ws_ln())


def test_gemm_split_k():
    _assert_all(gc.check_gemm_splitk())


@pytest.mark.parametrize("variant", ["reg", "glds", "naive"])
def test_conv(variant):
    _assert_all(gc.check_conv((variant,)))


def test_norms():
    _assert_all(gc.check_norms())



def test_attention():
    _assert_all(gc.check_attention())


def test_attention_small_mfma_clip_shapes():
    _assert_all(gc.check_attention_small_mfma())


def test_erf_gelu_on_every_finite_fp16_input():
    _assert_all(gc.check_gelu_all_inputs())


def test_elementwise():
    _assert_all(gc.check_elementwise())


def test_unet_vs_reference_generated_golden():
    _assert_all(gc.check_unet_golden())


def test_unet_mini_step_vs_oracle():
    # every comparison: HIP error vs the fp32 CPU oracle <= 2 x the error of the same oracle model run by PyTorch-ROCm eager
    # in fp16 on this GPU (SURVEY.md 8(c) tolerance policy)
    _assert_all(gc.check_unet_vs_oracle("mini", 3, 4, 8))
    _assert_all(gc.check_unet_vs_oracle("mini", 1, 8, 16, with_pnp=False))


def test_foreign_hook_code_drives_the_native_unet():
    """Seams B1 / B2 on the real kernels: torch-style processors + a replaced ResNet ``forward`` from outside the product
    package (``oracle.pnp_oracle``'s restatement of the reference hooks; the reference file itself is exercised by the CPU
    suite, /root/reference does not exist on the GPU box)."""
    _assert_all(gc.check_foreign_hooks("oracle"))


def test_pipeline_loops_vs_oracle_and_graph_equals_eager():
    _assert_all(gc.check_loops_mini())


def test_source_feature_cache_multi_edit_is_bit_equal():
    """several edits of one clip -- the source branch's injected features are recorded once and replayed."""
    _assert_all(gc.check_source_cache())


def test_two_branch_steps_bit_equal_at_a_mid_size_full_width():
    """at 16 f x 256^2 one branch has 16384 rows at the 320-channel level -- below the weight-stationary / LayerNorm-fold
    threshold on its own, above it when two branches' rows are scaled by the batch hint.  The fold decision is taken on the rows
    of ONE branch, so the replayed [negative, editing] steps stay bit-equal to the three-branch steps at this size too (the mini
    model has no 320-wide block and cannot see this)."""
    _assert_all(gc.check_source_cache("full", 16, 32, n_steps=4))
    gc.release_models()


def test_step_engines_are_reused_across_clips_without_stale_state():
    """HIP graphs captured for one clip replay on the next one (re-pointed static buffers): bit-equal to a fresh pipeline."""
    _assert_all(gc.check_engine_reuse_across_clips())


def test_unet_mini_long_clips():
    """BASELINE config 5 geometry in miniature: 128 frames (temporal attention through the flash kernel with a frame
    stride, S = 128 > 16) with PnP hooks, and a 40-frame clip."""
    _assert_all(gc.check_unet_vs_oracle("mini", 3, 128, 8))
    _assert_all(gc.check_unet_vs_oracle("mini", 1, 40, 16, with_pnp=False))
    gc.release_models()


# ---- the full 1.42 B-parameter model at the sizes that are benchmarked ------------------------------
def test_full_model_config1_B3_hooks_vs_cpu_oracle_and_reference_fixture():
    """BASELINE config 1 (1 clip x 8 f x 256x256), B=1 and B=3 with all 17 hook sites at t in {981, 301} -- the cpu_baseline
    workload -- vs the fp32 CPU oracle and vs the full-width fixture generated by the reference's own pnp_utils.py."""
    _assert_all(gc.check_n1_config1("full", 8, 32))


def test_full_model_config3_step_vs_fp32_oracle():
    """One step at the headline size [B,4,16,64,64], B=1 and B=3 + hooks, vs the fp32 oracle (torch-eager on the GPU)."""
    _assert_all(gc.check_n1_config3_step("full", 16, 64))


_LONG = __import__("os").environ.get("ANYV2V_LONG_TESTS", "0") == "1"


def test_full_model_config5_geometry_step_vs_fp32_oracle():
    """BASELINE config 5's geometry at FULL width: 128 frames (gradio_demo.py:129-131 / predict.py:153-155), so temporal
    attention S = 128 goes through the frame-strided flash kernel instead of the short-sequence one and a spatial launch sees
    N = 384 images; latent 16 x 16 here.  B=1 and B=3 with all 17 hook sites at t=981, bound 2 x the eager-fp16 oracle's error
    (VERDICT r2 missing #2); the eager oracles run through oracle/chunked.py (frame / pixel chunks, same arithmetic)."""
    res = gc.check_n1_config3_step("full", 128, 16, hooked_ts=(981,))
    for r in res:
        print(f"{'ok  ' if r['ok'] else 'FAIL'} {r['name']}: {r['err']:.3e} (tol {r['tol']:.2e})")
    gc.release_models()
    _assert_all(res)


def test_full_model_config5_midsize_pnp_step_vs_fp32_oracle():
    """Config 5 at 128 f x 32 x 32 latents (256^2 pixels), B=3 + 17 hook sites at t=981: four times the tokens of
    the geometry row above, every temporal layer at S = 128, N = 384 images per spatial launch at S = 1024 / 256 / 64 / 16 --
    HIP vs the chunked fp32 oracle, 2 x the eager-fp16 error, and |HIP - eager fp16| bounded by 3 x the B=1 row's eager error."""
    res = gc.check_n1_config3_step("full", 128, 32, hooked_ts=(981,))
    for r in res:
        print(f"{'ok  ' if r['ok'] else 'FAIL'} {r['name']}: {r['err']:.3e} (tol {r['tol']:.2e})")
    gc.release_models()
    _assert_all(res)


def test_full_model_config5_step_vs_fp32_oracle():
    """BASELINE config 5 at FULL width and FULL size: one PnP step at [3,4,128,64,64] with all 17 hook sites at t=981 (the path
    gradio_demo.py:129-131 / predict.py:153-155 drive) vs the fp32 oracle evaluated in chunks (oracle/chunked.py: no eager tensor
    beyond the config-3 rows' sizes; ~50 s of fp32 eager), bound min(2 x the eager-fp16 oracle's error, 3 x the recorded HIP error).
    Round 3's un-chunked checker sat 0.142 from BOTH fp16 paths at this size: MIOpen's fp32 3x3 convolution of the [384,960,64,64]
    skip-concat input is wrong (whole-vs-pieces max-rel 1.2, profiles/r04_eager_size_probe_fp32.txt); chunked, the HIP path is at
    2.1e-3 (profiles/r04_gputest_n1_log.txt).  ANYV2V_LONG_TESTS=1 adds the B=1 row; ANYV2V_CONFIG5_PLAIN_FP32=1 also runs the
    UN-chunked fp32 / fp16 eager oracles against the chunked ones (diagnostic, several GPU-minutes)."""
    import os
    res = gc.check_n1_config3_step("full", 128, 64, hooked_ts=(981,), plain_fp32_too=os.environ.get("ANYV2V_CONFIG5_PLAIN_FP32", "0") == "1",
                                   batches=(1, 3) if _LONG else (3,))
    for r in res:
        print(f"{'ok  ' if r['ok'] else 'FAIL'} {r['name']}: {r['err']:.3e} (tol {r['tol']:.2e})")
    gc.release_models()
    _assert_all(res)


@pytest.mark.skipif(not _LONG,
                    reason="~3 GPU-minutes: 500 fp32 + 500 fp16 oracle forwards at 16 f x 512^2; set ANYV2V_LONG_TESTS=1 "
                           "(last run: profiles/r03_gputest_n1_log.txt)")
def test_full_model_500_step_inversion_as_shipped():
    """configs/group_ddim_inversion/template.yaml:33 -- the reference's shipped 500-step inversion -- drift vs the fp32 oracle loop."""
    res = gc.check_inversion_500("full", 16, 64, n_steps=500, every=100)
    for r in res:
        print(f"{'ok  ' if r['ok'] else 'FAIL'} {r['name']}: {r['err']:.3e} (tol {r['tol']:.2e})")
    gc.release_models()
    _assert_all(res)


def test_full_model_50_step_drift_inversion_edit_reconstruction():
    """50-step inversion -> 50-step PnP edit and -> 50-step CFG reconstruction at 16 f x 512^2 through the product pipeline
    (HIP graphs) vs the oracle loops; final-latent drift <= 2 x the eager-fp16 oracle's drift, per-10-step drift reported."""
    res = gc.check_n1_drift("full", 16, 64, n_steps=50, every=10)
    for r in res:
        print(f"{'ok  ' if r['ok'] else 'FAIL'} {r['name']}: {r['err']:.3e} (tol {r['tol']:.2e})")
    gc.release_models()
    _assert_all(res)


def test_full_size_properties():
    """Size-independent identities at the BASELINE config 3 sizes (the oracle cannot run there in test time)."""
    _assert_all(gc.check_full_size_properties())


def test_vae_kernel_modes():
    _assert_all(gc.check_vae_kernels())


def test_native_vae_blocks_vs_reference_block_fixture():
    _assert_all(gc.check_vae_blocks_vs_reference_fixture())


def test_native_vae_vs_oracle():
    """AutoencoderKL encode / decode on the HIP kernels vs the CPU restatement (mini + full architecture)."""
    _assert_all(gc.check_vae())


def test_native_clip_towers_vs_transformers():
    """CLIP text (causal, 16 x 64) and vision (ViT-H/14, 16 x 80) towers on the HIP kernels vs transformers fp32."""
    pytest.importorskip("transformers")
    _assert_all(gc.check_clip())


def test_consisti2v_hook_family_vs_the_references_own_blocks_and_hooks():
    """ConsistI2V's decoder blocks + PnP hooks on the kernels vs a fixture produced by the reference's own code."""
    _assert_all(gc.check_consisti2v_hooks())


def test_consisti2v_whole_unet_vs_the_references_own_unet_and_hooks():
    """ConsistI2V end to end, UNet level: every block type of ``VideoLDMUNet3DConditionModel`` on the kernels vs the reference's own
    class (fixture), un-hooked and under its PnP hooks."""
    _assert_all(gc.check_consisti2v_unet())


def test_consisti2v_unet_at_the_released_width_vs_the_references_own_unet():
    """The 1250 M-parameter configuration at 16 f x 256^2 against a fixture of the reference's own class (fp32, CPU)."""
    res = gc.check_consisti2v_unet_full()
    for r in res:
        print(f"{'ok  ' if r['ok'] else 'FAIL'} {r['name']}: {r['err']:.3e} (tol {r['tol']:.2e})")
    _assert_all(res)


def test_consisti2v_pipeline_vs_the_references_own_pipeline_class():
    """ConsistI2V end to end, pipeline level: inversion, reconstruction and PnP edit on the kernels vs the reference's own
    ``ConditionalVideoEditingPipeline`` (fixture)."""
    _assert_all(gc.check_consisti2v_pipeline())


def test_unet_at_latent_sizes_that_are_not_multiples_of_8_vs_oracle():
    """Latent sizes that three ceil-halvings do not give back by doubling (``forward_upsample_size``): stride-2 convolutions on odd
    sizes, the gather + plain convolution on the way up, ragged attention lengths -- vs the fp32 oracle, PnP hooks on."""
    res = gc.check_unet_vs_oracle("mini", 1, 2, (9, 10), with_pnp=False) + gc.check_unet_vs_oracle("mini", 3, 2, (12, 9))
    _assert_all(res)


def test_consisti2v_samplers_and_options_vs_the_references_own_classes():
    """Both animation pipelines and ``guidance_rescale`` + ``eta`` on the kernels vs the reference's own classes (fixture)."""
    _assert_all(gc.check_consisti2v_sampling())


def test_seine_hook_family_vs_the_references_own_blocks_and_hooks():
    """SEINE's decoder blocks + PnP hooks (incl. the cross-attention hook) on the kernels vs a fixture produced by the
    reference's own code."""
    _assert_all(gc.check_seine_hooks())


def test_seine_whole_unet_vs_the_references_own_unet_and_hooks():
    _assert_all(gc.check_seine_unet())


def test_seine_runner_classes_vs_the_references_own_runner_classes():
    _assert_all(gc.check_seine_pipeline())


def test_attention_score_bias_and_per_head_rotary():
    _assert_all(gc.check_attention_bias_and_rotary_windows())


def test_pipeline_vs_the_reference_pipelines_own_output():
    """north_star: "outputs match the reference PyTorch CPU path on the same inputs".  The fixtures hold what the REFERENCE'S OWN
    PIPELINE CLASS (pipeline_i2vgen_xl.py, verbatim) produced on the CPU in fp32 around the oracle UNet -- inversion trajectory,
    CFG reconstruction, PnP edit -- for a mini UNet and for the full 1.42 B UNet at BASELINE config 1's size; the product pipeline
    (HIP kernels, HIP graphs) runs the same inputs.  Every bound: 2 x the eager-fp16 oracle's error on this GPU."""
    res = gc.check_pipeline_vs_reference_fixture("mini") + gc.check_pipeline_vs_reference_fixture("full")
    for r in res:
        print(f"{'ok  ' if r['ok'] else 'FAIL'} {r['name']}: {r['err']:.3e} (tol {r['tol']:.2e})")
    gc.release_models()
    _assert_all(res)


def test_graph_capture_is_safe_against_the_cyclic_garbage_collector():
    """A ``CUDAGraph`` freed by Python's cyclic GC while ANOTHER graph is being captured aborts the process on ROCm ("operation not
    permitted when stream is capturing", raised in the destructor; seen in the round-5 suite when a collection happened to run inside
    a step engine's capture, and torch >= 2.9 no longer collects before a capture).  ``utils.capture_hip_graph`` collects first and
    keeps the collector off during the capture: here an old graph becomes cyclic garbage INSIDE the capture region with the GC
    thresholds at 1, i.e. a collection would run at the next allocation."""
    import gc

    import torch

    from anyv2v_amd.utils import capture_hip_graph
    x = torch.ones(1024, device="cuda")
    y = torch.zeros(1024, device="cuda")

    def captured():
        g = torch.cuda.CUDAGraph()
        with capture_hip_graph(g):
            y.add_(x)
        return g

    old_graph = captured()

    class Holder:
        pass

    thresholds = gc.get_threshold()
    gc.set_threshold(1, 1, 1)
    try:
        g = torch.cuda.CUDAGraph()
        with capture_hip_graph(g):
            h = Holder()
            h.graph, h.me = old_graph, h        # a reference cycle that owns the old graph ...
            del h, old_graph                    # ... and is now unreachable
            junk = [[i] for i in range(2000)]   # allocations: with the collector on, it would run here, inside the capture
            y.add_(x)
            del junk
        assert gc.isenabled()
    finally:
        gc.set_threshold(*thresholds)
    gc.collect()                                # the old graph dies here, outside any capture
    y.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert float(y.sum()) == 1024.0
