"""Which GEMM launches of a batch-hinted two-branch forward plan a different split-K factor than the three-branch launch they stand
for?  Runs the full-width UNet once with B = 3 and once with B = 2 under ops.batch_hint(3, 2) (ANYV2V_GEMM_LOG=1 lines on stderr are
captured through a pipe) at a given geometry and compares the plans launch by launch.   python tools/hint_plan_diff.py [frames] [hw]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import types
    import torch
    import gpu_checks as gc
    from anyv2v_amd import ops, pnp_utils
    B, Fr, hw = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    m = gc.full_models("full", 1234, want=("native",))
    native, ocfg = m["native"], m["ocfg"]
    inp = gc.config1_inputs(ocfg, 3, Fr, hw)
    g = lambda x: x.half().to("cuda") if x.is_floating_point() else x.to("cuda")
    kw = dict(fps=g(inp["fps"]), image_latents=g(inp["image_latents"]), image_embeddings=g(inp["image_embeddings"]),
              encoder_hidden_states=g(inp["encoder_hidden_states"]))
    smp = g(inp["sample"])
    p2 = types.SimpleNamespace(unet=native)
    tsl = [981 - 20 * i for i in range(50)]
    pnp_utils.register_temp_attention_pnp(p2, tsl)       # temporal injection only (the state in which the mismatch showed)
    pnp_utils.register_time(p2, 981)
    native(smp[:1], 981, **{k: v[:1] for k, v in kw.items()})   # warm-up (conditioning cache etc.), not logged separately
    sys.stderr.write("=== START\n")
    if B == 3:
        native(smp, 981, **kw)
    else:
        with ops.batch_hint(3, 2):
            native(smp[1:], 981, **{k: v[1:] for k, v in kw.items()})
    torch.cuda.synchronize()
    sys.stderr.write("=== END\n")
    sys.exit(0)

Fr = int(sys.argv[1]) if len(sys.argv) > 1 else 16
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 32
logs = {}
for B in (3, 2):
    env = dict(os.environ, ANYV2V_GEMM_LOG="1", ANYV2V_FF_FUSED="0")
    r = subprocess.run([sys.executable, __file__, "--child", str(B), str(Fr), str(hw)], env=env, capture_output=True, text=True)
    lines = r.stderr.split("=== START\n")[-1].split("=== END")[0].splitlines()
    logs[B] = [l for l in lines if l.startswith("gemm-plan")]
    print(f"B={B}: {len(logs[B])} launches, {sum('splits 1' not in l for l in logs[B])} split")
import re
pat = re.compile(r"mode (\d) M (\d+) \(hinted (\d+)\) N (\d+) K (\d+) act (\d) res (\d) big (\d) splits (\d+)")
three = {}
for l in logs[3]:
    mode, M, Mh, N, K, act, res, big, sp = map(int, pat.search(l).groups())
    three.setdefault((mode, M, N, K, act, res), set()).add(sp)
bad = 0
for l in logs[2]:
    mode, M, Mh, N, K, act, res, big, sp = map(int, pat.search(l).groups())
    ref = three.get((mode, Mh, N, K, act, res))
    if ref is None:
        print("  no three-branch launch with", (mode, Mh, N, K, act, res), "for", l)
    elif sp not in ref:
        bad += 1
        print(f"  MISMATCH: two-branch {l}  |  three-branch splits {sorted(ref)}")
print("mismatching launches:", bad)
