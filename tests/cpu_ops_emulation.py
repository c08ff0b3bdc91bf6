"""TEST-ONLY emulation of the C-ABI ops (``anyv2v_amd.ops``) with plain torch on the CPU.

Purpose: exercise the *host logic* (UNet wiring over the token layout, weight packing, conditioning cache,
PnP aliasing, pipeline loops, graph-free step engine) in the ``-m "not gpu"`` suite, where no GPU exists.
It is installed by monkeypatching inside tests and is never importable from the product package: the product
path has no CPU fallback (``anyv2v_amd._lib`` raises when the HIP library is missing).
Each function restates the contract documented in ``include/anyv2v_hip.h``.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

MODE_LINEAR, MODE_CONV2D, MODE_TEMPORAL = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_GELU, ACT_GEGLU, ACT_F32OUT = 0, 1, 2, 3, 4


def _h(x):
    return x.to(torch.float16)


def gemm(a0, w, *, a1=None, bias=None, rowvec=None, rowvec_div=0, residual=None, act=ACT_NONE, out=None,
         mode=MODE_LINEAR, conv=None, temporal=None, M=None, naive=False, ln=None):
    a = a0.float() if a1 is None else torch.cat([a0.float(), a1.float()], 1)
    if ln is not None:   # LayerNorm fold: gamma / beta live in w / bias, the rows only need centring and scaling
        mean = a.mean(1, keepdim=True)
        a = (a - mean) * torch.rsqrt(a.var(1, unbiased=False, keepdim=True) + ln[1])
    K = a.shape[1]
    N = w.shape[0]
    wf = w.float()
    if mode == MODE_LINEAR:
        y = a @ wf.t()
    elif mode == MODE_CONV2D:
        Hi, Wi, Ho, Wo, stride, up = conv[:6]
        asym = conv[6] if len(conv) > 6 else 0
        n = a.shape[0] // (Hi * Wi)
        x = a.view(n, Hi, Wi, K).permute(0, 3, 1, 2)
        if up:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        w4 = wf.view(N, 3, 3, K).permute(0, 3, 1, 2)
        if asym:
            y = F.conv2d(F.pad(x, (0, 1, 0, 1)), w4, None, stride=stride, padding=0)
        else:
            y = F.conv2d(x, w4, None, stride=stride, padding=1)
        assert y.shape[2] == Ho and y.shape[3] == Wo
        y = y.permute(0, 2, 3, 1).reshape(n * Ho * Wo, N)
    else:
        Fr, HW = temporal
        B = a.shape[0] // (Fr * HW)
        x = a.view(B, Fr, HW, K).permute(0, 3, 1, 2).unsqueeze(-1)
        w5 = wf.view(N, 3, K).permute(0, 2, 1)[:, :, :, None, None]
        y = F.conv3d(x, w5, None, padding=(1, 0, 0)).squeeze(-1).permute(0, 2, 3, 1).reshape(B * Fr * HW, N)
    if M is not None:
        assert y.shape[0] == M, (y.shape, M)
    if act == ACT_F32OUT:
        if bias is not None:
            y = y + bias.float()
        if out is None:
            return y
        out[: y.shape[0], : y.shape[1]] = y
        return out
    if act == ACT_GEGLU:
        if bias is not None:
            y = y + bias.float()
        y = y.view(y.shape[0], N // 32, 2, 16)
        y = (y[:, :, 0] * F.gelu(y[:, :, 1])).reshape(y.shape[0], N // 2)
    else:
        if bias is not None:
            y = y + bias.float()
        if rowvec is not None:
            idx = torch.arange(y.shape[0]) // rowvec_div
            y = y + rowvec.float()[idx]
        if act == ACT_SILU:
            y = F.silu(y)
        elif act == ACT_GELU:
            y = F.gelu(y)
    if residual is not None:
        y = _h(y).float() + residual.float()
    y = _h(y)
    if out is None:
        return y
    out[: y.shape[0], : y.shape[1]] = y
    return out


def groupnorm(x0, gamma, beta, stats, rows_per_group, *, x1=None, groups=32, eps=1e-5, silu=False, out=None, shard=None):
    x = x0.float() if x1 is None else torch.cat([x0.float(), x1.float()], 1)
    M, C = x.shape
    n = M // rows_per_group
    if shard is not None:  # two-phase contract of anyv2v_groupnorm_partial_f16 / _apply_f16: sums -> all-reduce -> apply
        shards, all_reduce_sum = shard
        xg = x.view(n, rows_per_group, groups, C // groups)
        sums = torch.stack([xg.sum((1, 3)), (xg * xg).sum((1, 3))], -1).contiguous()  # [n, G, 2]
        buf = stats[: sums.numel()]
        buf.copy_(sums.reshape(-1))
        all_reduce_sum(buf)
        sums = buf.view(n, groups, 2)
        cnt = rows_per_group * (C // groups) * shards
        mean = sums[..., 0] / cnt
        rstd = torch.rsqrt((sums[..., 1] / cnt - mean * mean).clamp_min(0) + eps)
        y = ((xg - mean[:, None, :, None]) * rstd[:, None, :, None]).reshape(n, rows_per_group, C)
        y = (y * gamma.float() + beta.float()).permute(0, 2, 1)
    else:
        y = F.group_norm(x.view(n, rows_per_group, C).permute(0, 2, 1), groups, gamma.float(), beta.float(), eps)
    if silu:
        y = F.silu(y)
    y = _h(y.permute(0, 2, 1).reshape(M, C))
    if out is not None:
        out.copy_(y)
        return out
    return y


def layernorm(x, gamma, beta, eps=1e-5, out=None):
    y = _h(F.layer_norm(x.float(), (x.shape[1],), gamma.float(), beta.float(), eps))
    if out is not None:
        out.copy_(y)
        return out
    return y


def softmax_rows(s, scale, out=None):
    y = _h(torch.softmax(s.float() * scale, dim=-1))
    if out is not None:
        out.copy_(y)
        return out
    return y


def attention(q, k, v, out, *, batch, heads, Sq, Sk, inner=1, q_strides, kv_strides, kv_div=1, qk_mod=0, scale=0.125,
              head_dim=64, naive=False, causal=False, bias=None):
    i = torch.arange(batch)
    iq = i % qk_mod if qk_mod > 0 else i

    def rows(idx, st, S):
        base = (idx // inner) * st[0] + (idx % inner) * st[1]
        return base[:, None] + torch.arange(S)[None, :] * st[2]  # [batch, S]

    rq, ro = rows(iq, q_strides, Sq), rows(i, q_strides, Sq)
    rk, rv = rows(iq // kv_div, kv_strides, Sk), rows(i // kv_div, kv_strides, Sk)
    D = head_dim
    Q = q.float()[rq].view(batch, Sq, heads, D).transpose(1, 2)
    K = k.float()[rk].view(batch, Sk, heads, D).transpose(1, 2)
    V = v.float()[rv].view(batch, Sk, heads, D).transpose(1, 2)
    if bias is not None:
        O = F.scaled_dot_product_attention(Q, K, V, attn_mask=bias.float()[None], scale=scale)
    else:
        O = F.scaled_dot_product_attention(Q, K, V, scale=scale, is_causal=bool(causal))
    O = O.transpose(1, 2).reshape(batch, Sq, heads * D)
    out[ro.reshape(-1)] = _h(O.reshape(batch * Sq, heads * D))
    return out


def silu(x, out=None):
    y = _h(F.silu(x.float()))
    if out is not None:
        out.copy_(y)
        return out
    return y


def add(a, b, out=None):
    y = _h(a.float() + b.float())
    if out is not None:
        out.copy_(y)
        return out
    return y


def timestep_embedding(t_f32, dim, out=None):
    half = dim // 2
    fr = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    arg = t_f32.float()[:, None] * fr[None]
    return _h(torch.cat([arg.cos(), arg.sin()], -1))


def ncfhw_to_tokens(x, out, col0=0):
    B, C, Fr, H, W = x.shape
    out[:, col0:col0 + C] = x.permute(0, 2, 3, 4, 1).reshape(-1, C)
    return out


def tokens_to_ncfhw(x, B, Cc, Fr, H, W, col0=0, out=None):
    y = x[:, col0:col0 + Cc].reshape(B, Fr, H, W, Cc).permute(0, 4, 1, 2, 3).contiguous()
    if out is not None:
        out.copy_(y)
        return out
    return y


def adaptive_avgpool(x, N, Hi, Wi, Ho, Wo):
    C = x.shape[1]
    y = F.adaptive_avg_pool2d(x.float().view(N, Hi, Wi, C).permute(0, 3, 1, 2), (Ho, Wo))
    return _h(y.permute(0, 2, 3, 1).reshape(N * Ho * Wo, C))


def copy_cols(x, xcol0, y, ycol0, ncols):
    y[:, ycol0:ycol0 + ncols] = x[:, xcol0:xcol0 + ncols]
    return y


def gather_rows(x, xcol0, idx, y, ycol0, ncols):
    y[:, ycol0:ycol0 + ncols] = x[idx.long(), xcol0:xcol0 + ncols]
    return y


def rotary(x, col0, rot_dim, rows_per_pos, n_pos, theta=10000.0, windows=1, window_stride=0):
    rows = x.shape[0]
    pos = ((torch.arange(rows) // rows_per_pos) % n_pos).float()
    freq = theta ** (-torch.arange(0, rot_dim, 2).float() / rot_dim)
    ang = pos[:, None] * freq[None]
    cs, sn = ang.cos(), ang.sin()
    for w in range(windows):
        c = col0 + w * window_stride
        v = x[:, c:c + rot_dim].float()
        a, b = v[:, 0::2], v[:, 1::2]
        out = torch.stack([a * cs - b * sn, b * cs + a * sn], -1).reshape(rows, rot_dim)
        x[:, c:c + rot_dim] = _h(out)
    return x


def cfg_ddim_step(vtok, b_unc, b_cond, guidance, coef, lat, out):
    _, C, Fr, H, W = lat.shape
    n = Fr * H * W
    def branch(b):
        return vtok[b * n:(b + 1) * n, :C].float().view(Fr, H, W, C).permute(3, 0, 1, 2)
    v = branch(b_cond)
    if b_unc >= 0:
        vu = branch(b_unc)
        v = vu + guidance * (v - vu)
    sa_t, sb_t, sa_p, sb_p = [float(c) for c in coef[:4]]
    x = lat.float()[0]
    x0 = sa_t * x - sb_t * v
    eps = sa_t * v + sb_t * x
    out[0] = _h(sa_p * x0 + sb_p * eps)
    return out


def guided_step(e, x, coef, *, b_txt, b_unc=-1, b_img=-1, g_txt=1.0, g_img=1.0, prediction=0, out=None, noise=None, sigma=0.0):
    h = lambda t: t.half().float()
    eb = e.reshape(e.numel() // x.numel(), -1).float()
    v = eb[b_txt]
    if b_unc >= 0:
        if b_img >= 0:
            a = h(g_img * h(eb[b_img] - eb[b_unc]))
            b = h(g_txt * h(v - eb[b_img]))
            v = h(h(eb[b_unc] + a) + b)
        else:
            v = h(eb[b_unc] + h(g_txt * h(v - eb[b_unc])))
    sa_t, sb_t, sa_p, sb_p = (float(c) for c in coef)
    xf = x.reshape(-1).float()
    if prediction == 0:
        x0, eps = sa_t * xf - sb_t * v, sa_t * v + sb_t * xf
    elif prediction == 1:
        x0, eps = (xf - sb_t * v) / sa_t, v
    else:
        x0, eps = v, (xf - sa_t * v) / sb_t
    y = sa_p * x0 + sb_p * eps
    if noise is not None:
        y = y + float(sigma) * noise.reshape(-1).float()
    y = y.half().reshape(x.shape)
    if out is not None:
        out.copy_(y)
        return out
    return y


def ddim_step(v, x, sa_t, sb_t, sa_p, sb_p, out=None):
    v, x = v.float(), x.float()
    x0 = sa_t * x - sb_t * v
    eps = sa_t * v + sb_t * x
    y = _h(sa_p * x0 + sb_p * eps)
    if out is not None:
        out.copy_(y)
        return out
    return y


def ff_geglu(x, w1p, b1p, w2s, b2, residual=None, out=None):
    """include/anyv2v_hip.h, AnyV2VFFDesc: W1 / b1 rows interleaved [16 h | 16 gate] per 32; W2 slab-major [H / 32][C][32] with the
    slab's hidden units in MFMA slot order (column 8 q + e -> unit 4 q + e for e < 4, 16 + 4 q + e - 4 otherwise)."""
    T, Cc = x.shape
    H = w2s.shape[0] * 32
    proj = (x.float() @ w1p.float().t() + b1p.float()).view(T, H // 16, 2, 16)
    hidden = _h(proj[:, :, 0] * F.gelu(proj[:, :, 1])).float().reshape(T, H)      # one rounding, like the GEGLU GEMM's output
    q, e = torch.arange(4).view(4, 1), torch.arange(8).view(1, 8)
    perm = torch.where(e < 4, 4 * q + e, 16 + 4 * q + (e - 4)).reshape(32)
    w2 = torch.empty(Cc, H // 32, 32)
    w2[:, :, perm] = w2s.float().permute(1, 0, 2)                                 # slot order -> unit order
    y = _h(hidden @ w2.reshape(Cc, H).t() + b2.float())
    if residual is not None:
        y = _h(y.float() + residual.float())
    if out is not None:
        out.copy_(y)
        return out
    return y


def install(monkeypatch=None):
    """Replace every function of ``anyv2v_amd.ops`` with the emulation (tests only)."""
    from anyv2v_amd import ops
    names = ["gemm", "groupnorm", "layernorm", "softmax_rows", "attention", "silu", "add", "timestep_embedding", "ncfhw_to_tokens",
             "tokens_to_ncfhw", "adaptive_avgpool", "copy_cols", "gather_rows", "rotary", "cfg_ddim_step", "ddim_step", "guided_step", "ff_geglu"]
    g = globals()
    for n in names:
        if monkeypatch is not None:
            monkeypatch.setattr(ops, n, g[n])
        else:
            setattr(ops, n, g[n])
