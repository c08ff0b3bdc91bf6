"""C-ABI surface, config resolver and multi-process sharding (CPU; gloo world_size 2)."""
import ctypes
import json
import os
import re
import socket
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------- C ABI
def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "anyv2v_hip.h")).read()
    return sorted(set(re.findall(r"\b(anyv2v_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_symbol_the_header_declares():
    from anyv2v_amd import _lib
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    decl = _declared_symbols()
    assert len(decl) >= 17
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/anyv2v_hip.h but not exported"
    assert set(decl) == set(_lib.SYMBOLS), "ctypes binding table and header disagree"
    assert _lib.load().anyv2v_version() >= 103


def test_graft_entry_build_runs_here():
    """``__graft_entry__.build()`` is the driver's "does it build" check: make (a no-op when the objects are current) + the imports
    it lists.  (Round 5: it still imported a module that had been removed; nothing in the CPU suite called it while the .so existed.)"""
    import __graft_entry__ as g
    g.build()


def test_abi_argument_validation_without_gpu():
    """Host-side validation returns ANYV2V_EINVAL with a message before anything touches a device."""
    from anyv2v_amd import _lib
    lib = _lib.load()
    assert lib.anyv2v_gemm_f16(None, None) == -1
    assert b"null descriptor" in lib.anyv2v_last_error()
    d = _lib.GemmDesc()
    d.A0, d.W, d.C = 16, 16, 16
    d.M, d.N, d.C0, d.mode = 8, 8, 64, 7
    assert lib.anyv2v_gemm_f16(ctypes.byref(d), None) == -1
    assert b"bad mode" in lib.anyv2v_last_error()
    assert lib.anyv2v_layernorm_f16(None, None, None, None, 1, 1, 1e-5, None) == -1
    # sharded GroupNorm pair (frame-parallel clips): same validation as the one-call entry point, plus `shards`
    assert lib.anyv2v_groupnorm_partial_f16(None, None, 64, 0, None, 8, 8, 32, None) == -1
    assert b"null pointer" in lib.anyv2v_last_error()
    assert lib.anyv2v_groupnorm_partial_f16(16, None, 60, 0, 16, 8, 8, 32, None) == -1      # C0 % 8
    assert lib.anyv2v_groupnorm_partial_f16(16, None, 64, 0, 16, 9, 8, 32, None) == -1      # M % rows_per_group
    assert lib.anyv2v_groupnorm_apply_f16(16, None, 64, 0, 16, 16, 16, 16, 8, 8, 32, 1e-5, 0, 0, None) == -1
    assert b"shards" in lib.anyv2v_last_error()
    assert lib.anyv2v_groupnorm_partial_floats(8, 9, 32, 64) == -1 and lib.anyv2v_groupnorm_partial_floats(64, 8, 32, 64) == 8 * 32 * 2
    assert lib.anyv2v_groupnorm_partial_floats(64, 8, 32, 64) <= lib.anyv2v_groupnorm_scratch_floats(64, 8, 32)
    # round 3: LayerNorm fold only on the weight-stationary shapes; rotary / row gather / attention bias argument checks
    d = _lib.GemmDesc()
    d.A0, d.W, d.C = 16, 16, 16
    d.M, d.N, d.C0, d.lda0, d.ldc, d.flags = 64, 64, 64, 64, 64, 2
    d.ln_c1, d.ln_eps = 16, 1e-5
    assert lib.anyv2v_gemm_f16(ctypes.byref(d), None) != 0                # K = 64: no folded form
    assert lib.anyv2v_rotary_f16(None, 64, 8, 0, 32, 1, 0, 4, 2, 10000.0, None) == -1
    assert lib.anyv2v_rotary_f16(16, 64, 8, 0, 36, 1, 0, 4, 2, 10000.0, None) == -1           # rot_dim % 8
    assert lib.anyv2v_rotary_f16(16, 64, 8, 40, 32, 1, 0, 4, 2, 10000.0, None) == -1          # window leaves the row
    assert lib.anyv2v_rotary_f16(16, 64, 8, 0, 32, 2, 16, 4, 2, 10000.0, None) == -1          # overlapping windows
    assert b"rotary" in lib.anyv2v_last_error()
    assert lib.anyv2v_gather_rows_f16(16, 64, 0, None, 16, 64, 0, 8, 32, None) == -1          # null index
    assert lib.anyv2v_gather_rows_f16(16, 64, 4, 16, 16, 64, 0, 8, 32, None) == -1            # unaligned column window
    a = _lib.AttnDesc()
    assert lib.anyv2v_attention_bias_f16(ctypes.byref(a), 64, None, None) == -1 and b"null bias" in lib.anyv2v_last_error()
    assert lib.anyv2v_attention_bias_f16(ctypes.byref(a), 200, 16, None) == -1                 # head_dim > 160
    # round 4: fused feed-forward -- null pointers, unsupported width (the caller then runs the two GEMMs), alignment
    assert lib.anyv2v_ff_geglu_f16(None, None) == -1
    f = _lib.FFDesc()
    assert lib.anyv2v_ff_geglu_f16(ctypes.byref(f), None) == -1 and b"null X" in lib.anyv2v_last_error()
    f.X = f.W1 = f.b1 = f.W2 = f.b2 = f.Y = 16
    f.M, f.C, f.H, f.ldx, f.ldy = 64, 640, 2560, 640, 640
    assert lib.anyv2v_ff_geglu_f16(ctypes.byref(f), None) == -2 and b"only C = 320" in lib.anyv2v_last_error()
    f.C, f.H, f.ldx, f.ldy = 320, 1280, 324, 320
    assert lib.anyv2v_ff_geglu_f16(ctypes.byref(f), None) == -1 and b"ldx" in lib.anyv2v_last_error()
    # round 4: guidance + step (+ noise) -- null pointers, branch indices, prediction types that are singular at the given alpha
    assert lib.anyv2v_guided_step_f16(None, 8, -1, -1, 0, 1.0, 1.0, 1, 0.5, 0.5, 0.5, 0.5, 16, 16, None) == -1
    assert b"guided_step" in lib.anyv2v_last_error()
    assert lib.anyv2v_guided_step_f16(16, 8, -1, 1, 0, 1.0, 1.0, 1, 0.5, 0.5, 0.5, 0.5, 16, 16, None) == -1      # image branch without an unconditional one
    assert lib.anyv2v_guided_step_f16(16, 8, -1, -1, 0, 1.0, 1.0, 3, 0.5, 0.5, 0.5, 0.5, 16, 16, None) == -1     # unknown prediction type
    assert lib.anyv2v_guided_step_f16(16, 8, -1, -1, 0, 1.0, 1.0, 1, 0.0, 1.0, 0.5, 0.5, 16, 16, None) == -1     # epsilon at alpha = 0
    assert b"singular" in lib.anyv2v_last_error()
    assert lib.anyv2v_guided_step_noise_f16(16, 8, -1, -1, 0, 1.0, 1.0, 1, 0.5, 0.5, 0.5, 0.5, 16, 16, None, 0.3, None) == -1   # sigma without noise
    assert b"guided_step_noise" in lib.anyv2v_last_error()
    assert lib.anyv2v_set_batch_hint(0, 1) != 0 and lib.anyv2v_set_batch_hint(3, 2) == 0 and lib.anyv2v_set_batch_hint(1, 1) == 0
    with pytest.raises(_lib.HipKernelError):
        _lib.check(-1, "x")


def test_product_fails_loudly_without_the_hip_library(monkeypatch):
    from anyv2v_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libanyv2v_hip.so")
    with pytest.raises(_lib.HipExtensionMissing, match="no fallback"):
        _lib.load()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "anyv2v_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{fn} imports the oracle"
            assert "cpu_ops_emulation" not in src


# ------------------------------------------------------------------------------------------- config
def test_config_template_merge_and_interpolation(tmp_path):
    from anyv2v_amd.config import OmegaConf
    t = tmp_path / "template.yaml"
    t.write_text("""
seed: 8888
data_dir: ".."
model_name: "i2vgen-xl"
video_name: "ReplaceMe"
image_size: [512, 512]
n_frames: 16
output_dir: "${data_dir}/inversions/${model_name}/${video_name}"
inverse_config:
    image_size: ${image_size}
    n_frames: ${n_frames}
    n_steps: 500
    output_dir: "${output_dir}/ddim_latents"
recon_config:
    enable_recon: False
    ddim_latents_path: "${inverse_config.output_dir}"
""")
    tmpl = OmegaConf.load(str(t))
    cfg = OmegaConf.merge(tmpl, OmegaConf.create({"video_name": "Man Walking", "recon_config": {"enable_recon": True}}))
    assert cfg.output_dir == "../inversions/i2vgen-xl/Man Walking"
    assert cfg.inverse_config.output_dir == "../inversions/i2vgen-xl/Man Walking/ddim_latents"
    assert cfg.recon_config.ddim_latents_path == cfg.inverse_config.output_dir
    assert cfg.recon_config.enable_recon is True and cfg.inverse_config.n_steps == 500
    assert cfg.inverse_config.image_size == [512, 512] and cfg.inverse_config.n_frames == 16
    assert tmpl.video_name == "ReplaceMe"  # merge does not mutate the template
    cfg.video_path = os.path.join("x", cfg.video_name + ".mp4")  # attribute assignment like the runner
    assert cfg.video_path.endswith("Man Walking.mp4")
    assert "ReplaceMe" not in OmegaConf.to_yaml(cfg, resolve=True)
    with pytest.raises(AttributeError):
        _ = cfg.nope


def test_shipped_templates_resolve():
    from anyv2v_amd.config import OmegaConf
    base = os.path.join(ROOT, "configs")
    inv = OmegaConf.load(os.path.join(base, "group_ddim_inversion", "template.yaml"))
    e = json.load(open(os.path.join(base, "group_ddim_inversion", "group_config.json")))[0]
    c = OmegaConf.merge(inv, OmegaConf.create(e))
    assert c.inverse_config.output_dir.endswith(f"/inversions/i2vgen-xl/{c.video_name}/ddim_latents")
    assert c.inverse_config.cfg == 1.0 and c.inverse_config.prompt == "" and c.inverse_config.target_fps == 8
    pnp = OmegaConf.load(os.path.join(base, "group_pnp_edit", "template.yaml"))
    e = json.load(open(os.path.join(base, "group_pnp_edit", "group_config.json")))[0]
    c = OmegaConf.merge(pnp, OmegaConf.create(e))
    assert c.ddim_latents_path.endswith(f"/inversions/i2vgen-xl/{c.video_name}/ddim_latents")
    for k in ("seed", "device", "image_size", "n_frames", "cfg", "target_fps", "n_steps", "ddim_init_latents_t_idx",
              "ddim_inv_prompt", "random_ratio", "pnp_f_t", "pnp_spatial_attn_t", "pnp_temp_attn_t", "editing_prompt",
              "editing_negative_prompt", "edited_first_frame_path", "video_dir", "output_dir", "active"):
        assert k in c, k


def test_the_references_own_job_lists_resolve_entry_by_entry():
    """B5: the reference's shipped job lists (i2vgen-xl/configs/group_*/group_config.json: 15 edit entries, 6 inversion entries) are
    shipped with the same entries, and EVERY entry resolves through anyv2v_amd.config against the shipped template -- and, in this
    container, against the REFERENCE's own template.yaml -- to the paths and the output-directory suffix of
    run_group_pnp_edit.py:82-86,154-168 / run_group_ddim_inversion.py."""
    from anyv2v_amd.config import OmegaConf
    from anyv2v_amd.run_group_pnp_edit import output_suffix
    ours = os.path.join(ROOT, "configs")
    ref = "/root/reference/i2vgen-xl/configs"
    edits = json.load(open(os.path.join(ours, "group_pnp_edit", "group_config.json")))
    invs = json.load(open(os.path.join(ours, "group_ddim_inversion", "group_config.json")))
    assert len(edits) == 15 and len(invs) == 6
    assert sum(e["active"] is not False for e in edits) == 1 and edits[0]["active"] is True
    if os.path.isdir(ref):
        assert edits == json.load(open(os.path.join(ref, "group_pnp_edit", "group_config.json")))
        assert invs == json.load(open(os.path.join(ref, "group_ddim_inversion", "group_config.json")))
    bases = [ours] + ([ref] if os.path.isdir(ref) else [])
    resolved = {}
    for base in bases:
        tmpl = OmegaConf.load(os.path.join(base, "group_pnp_edit", "template.yaml"))
        for i, e in enumerate(edits):
            c = OmegaConf.merge(tmpl, OmegaConf.create(e))
            c.video_path = os.path.join(c.video_dir, c.video_name + ".mp4")
            c.video_frames_path = os.path.join(c.video_dir, c.video_name)
            c.edited_first_frame_path = os.path.join(c.data_dir, c.edited_first_frame_path)
            assert "ReplaceMe" not in OmegaConf.to_yaml(c, resolve=True), (base, i)
            assert c.ddim_latents_path == f"{c.data_dir}/inversions/i2vgen-xl/{e['video_name']}/ddim_latents"
            assert c.video_frames_path.endswith(os.path.join("demo", e["video_name"]))
            assert c.output_dir.rstrip("/") == f"{c.data_dir}/Results/{e['task_name']}/i2vgen-xl/{e['video_name']}/{e['edited_video_name']}"
            # the entry's overrides win over the template defaults (1 / 0.2 / 0.2 / 0.5); the suffix spells them as Python floats / ints
            t_idx = e.get("ddim_init_latents_t_idx", 1)
            f_t, s_t, t_t = e.get("pnp_f_t", 0.2), e.get("pnp_spatial_attn_t", 0.2), e.get("pnp_temp_attn_t", 0.5)
            assert (c.ddim_init_latents_t_idx, c.pnp_f_t, c.pnp_spatial_attn_t, c.pnp_temp_attn_t) == (t_idx, f_t, s_t, t_t)
            suffix = output_suffix(c, c.ddim_init_latents_t_idx)
            assert suffix == f"ddim_init_latents_t_idx_{t_idx}_nsteps_50_cfg_9.0_pnpf{f_t}_pnps{s_t}_pnpt{t_t}"
            resolved.setdefault(i, []).append((OmegaConf.to_yaml(c, resolve=True).replace("cuda:4", "cuda:0"), suffix))
        tmpl = OmegaConf.load(os.path.join(base, "group_ddim_inversion", "template.yaml"))
        for i, e in enumerate(invs):
            c = OmegaConf.merge(tmpl, OmegaConf.create(e))
            assert c.inverse_config.output_dir == f"{c.data_dir}/inversions/i2vgen-xl/{e['video_name']}/ddim_latents"
            assert c.recon_config.enable_recon is e.get("recon_config", {}).get("enable_recon", False)
            assert list(c.inverse_config.image_size) == e.get("image_size", [512, 512]) and c.inverse_config.n_steps == 500
    for i, both in resolved.items():   # our template and the reference's resolve every entry to the same values
        if len(both) == 2:
            a, b = (yaml_text for yaml_text, _ in both)
            strip = lambda t: sorted(l.split("#")[0].rstrip() for l in t.splitlines() if l.strip())
            assert strip(a) == strip(b), i


# ------------------------------------------------------------------------------------------- sharding
def test_shard_entries_round_robin_and_seeds():
    from anyv2v_amd.parallel import seed_for_entry, shard_entries
    entries = [{"active": i != 2, "id": i} for i in range(10)]
    parts = [shard_entries(entries, r, 4) for r in range(4)]
    ids = sorted(e["id"] for p in parts for e in p)
    assert ids == [0, 1, 3, 4, 5, 6, 7, 8, 9]
    assert [e["id"] for e in parts[0]] == [0, 5, 9] and max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert shard_entries(entries, 0, 1) == [e for e in entries if e["active"]]
    assert seed_for_entry(8888, 0) == 8888 and seed_for_entry(8888, 1) != seed_for_entry(8888, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from anyv2v_amd.parallel import gather_latents, init_distributed, shard_entries
    r, lr, w = init_distributed("gloo")
    entries = [{"active": i != 3, "id": i} for i in range(6)]   # 5 active entries: rank 0 runs three, rank 1 two
    mine = shard_entries(entries, r, w)
    lats = [torch.full((1, 4, 2, 3, 3), float(e["id"]), dtype=torch.float16) for e in mine]
    got = gather_latents(lats, 5, (1, 4, 2, 3, 3), torch.float16, "cpu")
    assert tuple(got.shape) == (5, 4, 2, 3, 3)
    torch.save([float(g[0, 0, 0, 0]) for g in got], os.path.join(out_dir, f"r{r}.pt"))
    none = gather_latents([], 1, (1, 4, 2, 3, 3), torch.float16, "cpu") if r == 1 else \
        gather_latents([torch.ones(1, 4, 2, 3, 3, dtype=torch.float16)], 1, (1, 4, 2, 3, 3), torch.float16, "cpu")
    assert float(none.sum()) == 4 * 2 * 3 * 3     # a rank without entries contributes an (ignored) zero slot
    dist.barrier()
    dist.destroy_process_group()


def _clipwise_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from anyv2v_amd.parallel import gather_latents, init_distributed, shard_by_clip
    r, lr, w = init_distributed("gloo")
    res = {}
    for tag, clips in (("AAB", "AAB"), ("AAAB", "AAAB"), ("ABA", "ABA")):
        inv = [{"video_name": "A"}, {"video_name": "B"}]
        edits = [{"video_name": c, "id": i} for i, c in enumerate(clips)]
        _, mine = shard_by_clip(inv, edits, r, w)
        idx = [e["id"] for e in mine]
        lats = [torch.full((1, 4, 2, 3, 3), float(i), dtype=torch.float16) for i in idx]
        got = gather_latents(lats, len(edits), (1, 4, 2, 3, 3), torch.float16, "cpu", indices=idx)
        res[tag] = [float(g[0, 0, 0, 0]) for g in got]
    # an entry held twice / not at all is refused on every rank together
    try:
        gather_latents([torch.zeros(1, 4, 2, 3, 3, dtype=torch.float16)], 2, (1, 4, 2, 3, 3), torch.float16, "cpu", indices=[0])
        res["dup"] = "no error"
    except ValueError as e:
        res["dup"] = "ValueError" if "exactly one" in str(e) else str(e)
    torch.save(res, os.path.join(out_dir, f"c{r}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_latents_scatters_by_entry_index_for_clipwise_dealing(tmp_path):
    """``shard_by_clip`` is not round-robin -- edits [A, A, B] put entries 0, 1 on rank 0 and 2 on rank 1, [A, A, A, B]
    three entries on rank 0 -- so the entry indices travel with the latents and the gathered tensor is scattered by index."""
    import torch.multiprocessing as mp
    mp.spawn(_clipwise_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "c0.pt"), torch.load(tmp_path / "c1.pt")
    assert a == b
    assert a["AAB"] == [0.0, 1.0, 2.0] and a["AAAB"] == [0.0, 1.0, 2.0, 3.0] and a["ABA"] == [0.0, 1.0, 2.0] and a["dup"] == "ValueError"


def test_all_gather_of_edited_latents_world2_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert a == b == [0.0, 1.0, 2.0, 4.0, 5.0]  # EVERY active entry's latents, in entry order, on every rank


def _bench_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo")
    lat = torch.full((1, 4, 2, 3, 3), float(rank + 1), dtype=torch.float16)
    dt, got = bench.finish_distributed(dist, 1.0 + rank, lat, world, "cpu")
    torch.save((dt, [float(g.mean()) for g in got]), os.path.join(out_dir, f"b{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_distributed_epilogue_world2_gloo(tmp_path):
    """bench.py's N>1 path: MAX-over-ranks timing + the one all_gather (RCCL on the node, gloo here)."""
    import torch.multiprocessing as mp
    mp.spawn(_bench_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        dt, means = torch.load(tmp_path / f"b{r}.pt")
        assert dt == 2.0 and means == [1.0, 2.0]


def test_attention_v_tile_swizzle_is_bank_conflict_free():
    """LDS bank model of MI355X (64 banks x 4 B; ds_read_b64_tr_b16 is serviced in two 32-lane groups; ds_read_b128 in
    four 16-lane groups) applied to the fragment addressing of the flash kernels: the V tile's chunk swizzle
    ``av_vswz`` must make every transpose read conflict-free (the previous ``key & 7`` cost 32 extra LDS cycles per KV
    tile -- measured by SQ_LDS_BANK_CONFLICT, predicted by this model), and the K tile's ``(key >> 1) & 7`` likewise."""
    import re
    src = open(os.path.join(ROOT, "anyv2v_amd", "csrc", "attention.hip")).read()
    m = re.search(r"int av_vswz\(int key\) \{ return ([^;]+); \}", src)
    assert m, "av_vswz not found"
    expr = m.group(1)
    swz = lambda key: eval(expr, {"key": key})  # noqa: S307  (expression of & | << only, from our own source)
    assert re.fullmatch(r"[\sk\(\)ey&|<0-9]+", expr)

    def extra_cycles(addr, groups, dwords):
        extra = 0
        for grp in groups:
            banks = {}
            for lane in grp:
                a = addr(lane) // 4
                for d in range(dwords):
                    banks.setdefault((a + d) % 64, set()).add(a + d)
            extra += max(len(v) for v in banks.values()) - 1
        return extra

    def v_addr(lane, db, t, second, f):
        hi, i16 = lane >> 5, lane & 15
        vrow = 4 * hi + (i16 >> 2)
        vc0 = 2 * ((lane >> 4) & 1) + ((i16 & 3) >> 1)
        return (vrow + 16 * t + 8 * second) * 128 + (((4 * db + vc0) ^ f(vrow)) << 4) + (i16 & 1) * 8

    halves = [range(0, 32), range(32, 64)]
    total = lambda f: sum(extra_cycles(lambda l: v_addr(l, db, t, s2, f), halves, 2)
                          for db in (0, 1) for t in range(4) for s2 in (0, 1))
    assert total(swz) == 0
    assert total(lambda key: key & 7) == 32  # the layout this replaced

    g16 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    g16 += [[x + 32 for x in g] for g in g16]
    k_addr = lambda lane, ks, kb: (32 * kb + (lane & 31)) * 128 + (((2 * ks + (lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4)
    assert sum(extra_cycles(lambda l: k_addr(l, ks, kb), g16, 4) for kb in (0, 1) for ks in range(4)) == 0


_RCCL_WORLD1 = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from anyv2v_amd.parallel import FrameParallel, gather_latents
import bench
lats = [torch.full((1, 4, 2, 3, 3), float(i), dtype=torch.float16, device=dev) for i in (2, 0, 1)]
got = gather_latents(lats, 3, (1, 4, 2, 3, 3), torch.float16, dev, indices=[2, 0, 1])      # held out of order: scattered by index
assert got.is_cuda and [float(g[0, 0, 0, 0]) for g in got] == [0.0, 1.0, 2.0], got[:, 0, 0, 0, 0]
dt, out = bench.finish_distributed(dist, 1.5, lats[0].contiguous(), 1, dev)
assert dt == 1.5 and len(out) == 1 and torch.equal(out[0], lats[0])
fp = FrameParallel()
x = torch.randn(2 * 4 * 6, 8, device=dev).half()                                            # B = 2, 4 local frames, HW = 6
y = fp.pixels_to_frames(fp.frames_to_pixels(x, 2, 4, 6), 2, 4, 6)
assert torch.equal(x, y)
s = torch.ones(5, device=dev)
fp.all_reduce_sum(s)
assert float(s.sum()) == 5.0
dist.barrier()
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
'''


@pytest.mark.gpu
def test_collectives_through_rccl_world_size_1():
    """The N > 1 code paths on the backend they run on at round end ("nccl" = RCCL): a one-rank process group on the GPU box drives
    ``gather_latents`` (device tensors, index scatter), ``bench.finish_distributed`` and the frame-parallel all-to-all / all-reduce
    wrappers through RCCL itself.  (No multi-GPU lease exists here; rank counts > 1 are covered on gloo.)"""
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_WORLD1, ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
