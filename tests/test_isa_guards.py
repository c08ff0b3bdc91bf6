"""Static guards on the generated gfx950 code (no GPU needed: hipcc cross-compiles).

The K-tiles of the GEMM kernels and the QK / PV phases of the attention kernel issue their LDS fragment reads as inline asm and wait
for them with hand-counted ``s_waitcnt lgkmcnt(n)`` (hipcc would drain to ``lgkmcnt(0)`` while an LDS-DMA load is in flight, see
``tools/wait_probe.hip``).  hipcc does not know that such a destination register is not valid yet: if it ever decided to spill or
copy one between the read and the wait, the kernel would compute on garbage (seen once, in an experiment with asm *global* loads at
256 VGPRs).  This test compiles the sources to assembly and fails if any scratch access or VGPR copy of a pending destination
sits in that window, or if an MFMA consumes a fragment that the preceding counted wait does not cover (LDS operations of a wave
retire in issue order, so ``lgkmcnt(n)`` leaves exactly the youngest n outstanding)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _asm(src, extra=()):
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-Wno-unused-result", *extra, "-S",
                          "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "anyv2v_amd", "csrc", src), "-o", "-"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


def _regs(operand):
    m = re.match(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", operand)
    return {int(m.group(1))} if m else set()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("src,extra", [("gemm.hip", ()), ("attention.hip", ("-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize")),
                                       ("ff_fused.hip", ("-fno-slp-vectorize",)), ("gemm_sw.hip", ()), ("gemm_swh.hip", ())])
def test_no_spill_or_copy_of_a_pending_asm_lds_read(src, extra):
    asm = _asm(src, extra)
    kernels = re.findall(r"^(_Z\w+):.*?s_endpgm", asm, re.S | re.M)
    assert kernels
    checked = 0
    for m in re.finditer(r"^(_Z\w+):(.*?)s_endpgm", asm, re.S | re.M):
        queue = []                # LDS operations in issue order: destination VGPRs of inline-asm reads (empty set for the rest)
        in_asm = False
        for line in m.group(2).split("\n"):
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith(";"):
                continue
            op, _, rest = t.partition(" ")
            args = [a.strip() for a in rest.split(",")]
            if op.startswith("ds_"):
                queue.append(_regs(args[0]) if (in_asm and op.startswith("ds_read")) else set())
                checked += in_asm and op.startswith("ds_read")
                continue
            if op == "s_waitcnt":
                w = re.search(r"lgkmcnt\((\d+)\)", rest)
                if w:             # LDS operations retire in order: all but the youngest n are complete
                    n = int(w.group(1))
                    queue = queue[len(queue) - n:] if n else []
                continue
            if op in ("s_barrier", "s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccz", "s_cbranch_vccnz", "s_cbranch_execz",
                      "s_cbranch_execnz", "s_branch") or t.endswith(":"):
                continue
            pending = set().union(*queue) if queue else set()
            if not pending:
                continue
            assert not op.startswith("scratch_"), f"{m.group(1)}: scratch access while asm LDS reads are pending: {t}"
            if op.startswith("v_mfma"):   # the hand-counted wait in front of this MFMA must cover its fragments
                used = set().union(*[_regs(a) for a in args[1:3]])
                assert not (used & pending), f"{m.group(1)}: MFMA reads a fragment whose asm LDS read is not waited for: {t}"
            if op.startswith("v_mov") or op.startswith("v_accvgpr_write"):
                src_regs = set().union(*[_regs(a) for a in args[1:]]) if len(args) > 1 else set()
                assert not (src_regs & pending), f"{m.group(1)}: copy of a pending asm LDS-read destination: {t}"
    assert checked > 50, f"only {checked} inline-asm LDS reads found in {src}: did the kernels change form?"
