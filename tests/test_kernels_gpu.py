"""Kernel-level numerics on a real MI355X: every C-ABI entry point against a plain
PyTorch fp32 evaluation of the same op on the same inputs (bf16-rounded where the
kernel consumes bf16).  Tolerances are stated per test."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.to(torch.bfloat16)


def _rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("tile", [64, 128, 256])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 384, 256), (300, 208, 128), (1024, 768, 512)])
def test_gemm_plain(gpu, tile, M, N, K):
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
    ref = a.float() @ w.float().T
    out32 = ops.gemm_bf16(a, w, out_dtype=torch.float32, tile=tile)
    torch.cuda.synchronize()
    # fp32 accumulation of exact bf16 products: only summation order differs
    assert _rel(out32, ref) < 2e-6
    out16 = ops.gemm_bf16(a, w, out_dtype=torch.bfloat16, tile=tile)
    assert torch.equal(out16, _bf(out32))  # bf16 output is the RNE rounding of the fp32 result


@pytest.mark.parametrize("K", [64, 128, 448])
def test_gemm_w4_matches_8wave(gpu, K):
    """The 4-wave 256x256 kernel (tile code 257: whole tiles, one K source) accumulates every output element in the
    same order as the 8-wave kernel, so the two are bit-identical on every epilogue; odd / tiny K-tile counts exercise
    the pipeline prologue and tail.  Plain fp32 result also against torch."""
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator(device="cpu").manual_seed(K)
    M, N = 512, 768
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    for kw in (dict(), dict(residual=res), dict(bias=bias, relu=True, residual=res), dict(silu_mul=True)):
        for dt in (torch.float32, torch.bfloat16):
            r8 = ops.gemm_bf16(a, w, out_dtype=dt, tile=256, **kw)
            for code in (257, 271, 272):  # 256x256 (2x2 waves), 256x192 (4x1 waves; N = 768 = 4 x 192), two-barrier deep-prefetch form
                if kw.get("silu_mul") and dt == torch.float32:
                    # the 4-wave kernel's SiLU epilogue exists for the 16-bit operand type only (16-byte stores): refused
                    with pytest.raises(Exception):
                        ops.gemm_bf16(a, w, out_dtype=dt, tile=code, **kw)
                    continue
                r4 = ops.gemm_bf16(a, w, out_dtype=dt, tile=code, **kw)
                assert torch.equal(r8, r4), (code, sorted(kw), dt)
    # fused q|k|v form: RoPE epilogue (bf16 out) with and without the LoRA second K source
    pos = torch.arange(128, dtype=torch.float32)
    ang = pos[:, None] * (1.0 / (10000.0 ** (torch.arange(0, 64, 2, dtype=torch.float32) / 64)))[None, :]
    cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
    a2 = _bf(torch.randn(M, 64, generator=g)).to(dev)
    w2 = _bf(torch.randn(N, 64, generator=g) * 0.05).to(dev)
    for kw in (dict(rope=(cos, sin, 512)), dict(rope=(cos, sin, 512), a2=a2, w2=w2)):
        r8 = ops.gemm_bf16(a, w, out_dtype=torch.bfloat16, tile=256, **kw)
        for code in (257, 271):
            r4 = ops.gemm_bf16(a, w, out_dtype=torch.bfloat16, tile=code, **kw)
            assert torch.equal(r8, r4), (code, sorted(kw))
    out = ops.gemm_bf16(a, w, out_dtype=torch.float32, tile=257)
    assert _rel(out, a.float() @ w.float().T) < 2e-6
    h = res.clone()
    ops.gemm_bf16(a, w, out=h, residual=h, tile=257)  # in place on the residual stream, as the decoder uses it
    assert torch.equal(h, ops.gemm_bf16(a, w, out_dtype=torch.float32, residual=res, tile=256))
    with pytest.raises(Exception):
        ops.gemm_bf16(a[:300], w, out_dtype=torch.float32, tile=257)  # not whole tiles -> refused, not mis-computed


def test_gemm_asymmetric_identity(gpu):
    """A = I against an asymmetric W catches a transposed C write."""
    from tcavt_amd import ops

    dev = gpu["device"]
    K = N = 128
    a = _bf(torch.eye(K)).to(dev)
    w = _bf(torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251).to(dev)
    out = ops.gemm_bf16(a, w, out_dtype=torch.float32, tile=128)
    assert torch.equal(out, w.float().T.contiguous())


@pytest.mark.parametrize("tile", [64, 128, 256])
def test_gemm_bias_relu_residual_dual(gpu, tile):
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator(device="cpu").manual_seed(5)
    M, N, K, K2 = 520, 256, 192, 64
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.1).to(dev)
    a2 = _bf(torch.randn(M, K2, generator=g)).to(dev)
    w2 = _bf(torch.randn(N, K2, generator=g) * 0.1).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    ref = torch.relu(a.float() @ w.float().T + a2.float() @ w2.float().T + bias) + res
    out = ops.gemm_bf16(a, w, out_dtype=torch.float32, bias=bias, relu=True, residual=res, a2=a2, w2=w2, tile=tile)
    assert _rel(out, ref) < 2e-6
    # in-place residual (C aliases residual), as the decoder's h += ... uses it
    h = res.clone()
    ops.gemm_bf16(a, w, out=h, residual=h, tile=tile)
    assert _rel(h, a.float() @ w.float().T + res) < 2e-6


@pytest.mark.parametrize("tile", [64, 128, 256])
def test_gemm_silu_mul(gpu, tile):
    from tcavt_amd import ops
    from tcavt_amd.layout import interleave_gate_up

    dev = gpu["device"]
    g = torch.Generator(device="cpu").manual_seed(6)
    M, I, K = 384, 512, 128
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    wg = _bf(torch.randn(I, K, generator=g) * 0.2)
    wu = _bf(torch.randn(I, K, generator=g) * 0.2)
    wgu = interleave_gate_up(wg, wu).to(dev)
    gate = a.float() @ wg.to(dev).float().T
    up = a.float() @ wu.to(dev).float().T
    ref = torch.nn.functional.silu(gate) * up
    out = ops.gemm_bf16(a, wgu, out_dtype=torch.float32, silu_mul=True, tile=tile)
    assert out.shape == (M, I)
    assert _rel(out, ref) < 5e-6


@pytest.mark.parametrize("tile", [128, 256])
def test_gemm_rope(gpu, tile):
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator(device="cpu").manual_seed(7)
    B, L, K = 3, 40, 128
    nq, nkv = 4, 1
    N = (nq + 2 * nkv) * 64
    a = _bf(torch.randn(B * L, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.1).to(dev)
    inv = 1.0 / (10000.0 ** (torch.arange(0, 64, 2).float() / 64))
    ang = torch.arange(L).float()[:, None] * inv[None, :]
    cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
    raw = (a.float() @ w.float().T).view(B, L, N // 64, 64)
    x1, x2 = raw[..., :32], raw[..., 32:]
    c, s = cos[None, :, None, :], sin[None, :, None, :]
    rot = torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], dim=-1)
    ref = raw.clone()
    ref[:, :, : nq + nkv] = rot[:, :, : nq + nkv]
    out = ops.gemm_bf16(a, w, out_dtype=torch.float32, rope=(cos, sin, (nq + nkv) * 64), tile=tile)
    assert _rel(out, ref.reshape(B * L, N)) < 2e-6


@pytest.mark.parametrize("H", [256, 2048])
def test_rmsnorm(gpu, H):
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator(device="cpu").manual_seed(H)
    x = (torch.randn(77, H, generator=g) * 3).to(dev)
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(dev)
    ob = torch.empty(77, H, dtype=torch.bfloat16, device=dev)
    of = torch.empty(77, H, dtype=torch.float32, device=dev)
    ops.rmsnorm(x, gamma, 1e-5, ob, of)
    ref = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * gamma
    assert _rel(of, ref) < 1e-6
    assert torch.equal(ob, _bf(of))


@pytest.mark.parametrize("D", [64, 768])
def test_layernorm_residual(gpu, D):
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator(device="cpu").manual_seed(D)
    x = torch.randn(50, D, generator=g).to(dev)
    r = torch.randn(50, D, generator=g).to(dev)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(dev)
    beta = (0.1 * torch.randn(D, generator=g)).to(dev)
    of = torch.empty_like(x)
    ob = torch.empty(50, D, dtype=torch.bfloat16, device=dev)
    ops.layernorm(x, gamma, beta, 1e-5, residual=r, out_f32=of, out_bf16=ob)
    ref = torch.nn.functional.layer_norm(x + r, (D,), gamma, beta, 1e-5)
    assert (of - ref).abs().max().item() < 5e-6
    assert torch.equal(ob, _bf(of))


def test_cast_and_embed(gpu):
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(1003, generator=g).to(dev)
    assert torch.equal(ops.cast_bf16(x), _bf(x))
    B, Nq, Lt, H, V = 3, 4, 9, 256, 50
    table = _bf(torch.randn(V, H, generator=g)).to(dev)
    ids = torch.randint(0, V, (B, Lt), generator=g).to(dev)
    img = torch.randn(B, Nq, H, generator=g).to(dev)
    vm, tm = torch.randn(H, generator=g).to(dev), torch.randn(H, generator=g).to(dev)
    h = torch.empty(B, Nq + Lt, H, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.embed_fuse(table, ids, img, vm, tm, h, flag)
    ref = torch.cat([img + vm, table[ids].float() + tm], dim=1)
    assert torch.equal(h, ref)
    assert flag.item() == 0
    ids[0, 0] = V + 5
    ops.embed_fuse(table, ids, img, vm, tm, h, flag)
    assert flag.item() == 1


def _attn_ref(qkv, B, L, nq, nkv, kv_len, scale):
    x = qkv.float().view(B, L, nq + 2 * nkv, 64)
    q = x[:, :, :nq].permute(0, 2, 1, 3)
    k = x[:, :, nq:nq + nkv].permute(0, 2, 1, 3).repeat_interleave(nq // nkv, dim=1)
    v = x[:, :, nq + nkv:].permute(0, 2, 1, 3).repeat_interleave(nq // nkv, dim=1)
    s = (q @ k.transpose(-1, -2)) * scale
    i = torch.arange(L, device=qkv.device)
    mask = (i[None, :] <= i[:, None])[None] & (i[None, None, :] < kv_len[:, None, None])
    s = s.masked_fill(~mask[:, None], float("-inf"))
    p = torch.softmax(s, dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B * L, nq * 64)


@pytest.mark.parametrize("B,L,nq,nkv", [(2, 64, 4, 1), (3, 256, 8, 2), (2, 200, 4, 2), (1, 528, 4, 1)])
def test_attn_causal_gqa(gpu, B, L, nq, nkv):
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator(device="cpu").manual_seed(L + nq)
    qkv = _bf(torch.randn(B * L, (nq + 2 * nkv) * 64, generator=g)).to(dev)
    kv_len = torch.tensor([L, max(17, L // 2), 33][:B], dtype=torch.int32, device=dev)
    out = torch.empty(B * L, nq * 64, dtype=torch.bfloat16, device=dev)
    ops.attn_causal_gqa(qkv, out, kv_len, B, L, nq, nkv, 0.125)
    ref = _attn_ref(qkv, B, L, nq, nkv, kv_len, 0.125)
    # P is rounded to bf16 before the PV product and the output is bf16: 2^-9 relative per element
    assert _rel(out, ref) < 4e-3
    assert (out.float() - ref).abs().max().item() < 0.03


def test_attn_spiked_row(gpu):
    """Force the online-softmax rescale: one late key dominates one query row."""
    from tcavt_amd import ops

    dev = gpu["device"]
    B, L, nq, nkv = 1, 128, 4, 1
    g = torch.Generator(device="cpu").manual_seed(11)
    x = torch.randn(B * L, (nq + 2 * nkv) * 64, generator=g) * 0.5
    x[100, 0:64] = 4.0            # query 100, head 0
    x[90, nq * 64:(nq + 1) * 64] = 4.0  # key 90 aligned with it (tile 2, after tiles 0/1 were accumulated)
    qkv = _bf(x).to(dev)
    kv_len = torch.tensor([L], dtype=torch.int32, device=dev)
    out = torch.empty(B * L, nq * 64, dtype=torch.bfloat16, device=dev)
    ops.attn_causal_gqa(qkv, out, kv_len, B, L, nq, nkv, 0.125)
    ref = _attn_ref(qkv, B, L, nq, nkv, kv_len, 0.125)
    assert (out.float() - ref).abs().max().item() < 0.03


@pytest.mark.parametrize("Lq,Lk,nh,dh,dtype", [(18, 18, 8, 96, torch.float32), (64, 64, 4, 16, torch.float32),
                                                (30, 256, 2, 1024, torch.bfloat16), (16, 6, 8, 96, torch.float32)])
def test_mha_small(gpu, Lq, Lk, nh, dh, dtype):
    from tcavt_amd import ops

    dev = gpu["device"]
    B = 3
    g = torch.Generator(device="cpu").manual_seed(Lq * Lk + dh)
    E = nh * dh
    q = (torch.randn(B, Lq, E, generator=g)).to(dtype).to(dev)
    k = (torch.randn(B, Lk, E, generator=g)).to(dtype).to(dev)
    v = (torch.randn(B, Lk, E, generator=g)).to(dtype).to(dev)
    key_len = torch.tensor([Lk, max(1, Lk // 2), max(1, Lk - 1)], dtype=torch.int32, device=dev)
    out = torch.empty(B, Lq, E, dtype=torch.float32, device=dev)
    scale = 1.0 / math.sqrt(dh)
    ops.mha(q, k, v, out, B, Lq, Lk, nh, dh, scale, key_len=key_len)
    qh = q.float().view(B, Lq, nh, dh).transpose(1, 2)
    kh = k.float().view(B, Lk, nh, dh).transpose(1, 2)
    vh = v.float().view(B, Lk, nh, dh).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) * scale
    j = torch.arange(Lk, device=dev)
    s = s.masked_fill((j[None, :] >= key_len[:, None])[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Lq, E)
    assert _rel(out, ref) < 1e-5


@pytest.mark.parametrize("M,N,K", [(2048, 192, 64), (100, 64, 2048), (33, 70, 18), (960, 2, 64)])
def test_gemm_f32(gpu, M, N, K):
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dev)
    w = torch.randn(N, K, generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    out = ops.gemm_f32(a, w, bias=bias, relu=True, residual=res)
    ref = torch.relu(a.double() @ w.double().T + bias.double()) + res.double()
    assert _rel(out, ref.float()) < 1e-6


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_batched_two_level_strides(gpu, dtype):
    """Per-(sample, head) products with outer/inner strides, as the cross-attention issues them."""
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator(device="cpu").manual_seed(21)
    B, nh, To, L, dh = 3, 2, 30, 96, 128
    H = nh * dh
    Lp = 128
    q = torch.randn(B * To, H, generator=g).to(dtype).to(dev)
    k = torch.zeros(B * L + 64, H, dtype=dtype, device=dev)
    k[: B * L] = torch.randn(B * L, H, generator=g).to(dtype).to(dev)
    S = torch.full((B * nh * To, Lp), float("nan"), device=dev)
    ops.gemm_batched(q, k, S, M=To, N=Lp, K=dh, lda=H, ldw=H, ldc=Lp, batch=B * nh, inner=nh, sA=(To * H, dh),
                     sW=(L * H, dh), sC=(nh * To * Lp, To * Lp), acc_scale=0.25)
    qf = q.float().view(B, To, nh, dh).permute(0, 2, 1, 3)
    kf = k[: B * L].float().view(B, L, nh, dh).permute(0, 2, 1, 3)
    ref = 0.25 * (qf @ kf.transpose(-1, -2))  # [B,nh,To,L]
    got = S.view(B, nh, To, Lp)[..., :L]
    assert _rel(got, ref) < 2e-6


def test_gemm_swapped_roles_bias_row_f16_out(gpu):
    """V^T = W_v . X^T + b_v[:, None] per sample, fp16 output (transposed projection)."""
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator(device="cpu").manual_seed(22)
    B, L, H, Lp = 3, 40, 128, 64
    x = torch.zeros(B * L + 64, H, dtype=torch.bfloat16, device=dev)
    x[: B * L] = _bf(torch.randn(B * L, H, generator=g)).to(dev)
    wv = _bf(torch.randn(H, H, generator=g) * 0.1).to(dev)
    bv = torch.randn(H, generator=g).to(dev)
    vT = torch.zeros(H, B * Lp, dtype=torch.float16, device=dev)
    ops.gemm_batched(wv, x, vT, M=H, N=Lp, K=H, lda=H, ldw=H, ldc=B * Lp, batch=B, inner=1, sA=(0, 0), sW=(L * H, 0),
                     sC=(Lp, 0), bias_row=bv)
    ref = (x[: B * L].float() @ wv.float().T + bv).view(B, L, H).permute(2, 0, 1)  # [H,B,L]
    got = vT.view(H, B, Lp)[:, :, :L].float()
    assert (got - ref.half().float()).abs().max().item() <= 2e-3 * ref.abs().max().item()
    assert _rel(got, ref) < 5e-4


def test_softmax_rows(gpu):
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator(device="cpu").manual_seed(23)
    rows, L, Lp = 50, 100, 128
    S = (torch.randn(rows, Lp, generator=g) * 3).to(dev)
    P = torch.full((rows, Lp), 7.0, dtype=torch.float16, device=dev)
    ops.softmax_rows(S, P, rows, L, Lp, Lp, Lp)
    ref = torch.softmax(S[:, :L], dim=-1)
    assert (P[:, L:] == 0).all()
    assert (P[:, :L].float() - ref).abs().max().item() < 1e-3


@pytest.mark.parametrize("M,N,K", [(2048, 64, 2048), (64, 2048, 2048), (40, 70, 1000)])
def test_gemm_f32_split_k_and_strided(gpu, M, N, K):
    """Few output tiles + long K takes the split-K path (float atomics); also the transposed-operand form
    gW = gy^T x used by the backward."""
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator(device="cpu").manual_seed(M + N + K + 1)
    a = torch.randn(M, K, generator=g).to(dev)
    w = torch.randn(N, K, generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    out = ops.gemm_f32(a, w, bias=bias, residual=res)
    ref = (a.double() @ w.double().T + bias.double() + res.double()).float()
    assert _rel(out, ref) < 1e-6
    # gW[n][k] = sum_m gy[m][n] x[m][k]
    gy = torch.randn(K, M, generator=g).to(dev)   # [rows=K (contraction), cols=M]
    x = torch.randn(K, N, generator=g).to(dev)
    gw = torch.empty(M, N, device=dev)
    ops.gemm_f32_strided(gy, 1, gy.stride(0), x, 1, x.stride(0), gw, M, N, K)
    assert _rel(gw, (gy.double().T @ x.double()).float()) < 1e-6


def test_gemm_w4_persistent_stream(gpu):
    """More 256x256 tiles than CUs: the 4-wave kernel runs one persistent workgroup per CU that walks its tiles as one
    K-tile stream (look-ahead continues into the next output tile).  288 tiles, uneven tiles per workgroup, 1..3 K-tiles:
    bit-identical to the one-workgroup-per-tile 8-wave kernel."""
    from tcavt_amd import ops

    dev = gpu["device"]
    M, N = 4096, 4608
    for K in (64, 192, 320):
        g = torch.Generator(device="cpu").manual_seed(K)
        a = _bf(torch.randn(M, K, generator=g)).to(dev)
        w = _bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
        res = torch.randn(M, N, generator=g).to(dev)
        for kw, dt in ((dict(), torch.bfloat16), (dict(residual=res), torch.float32), (dict(silu_mul=True), torch.bfloat16)):
            r8 = ops.gemm_bf16(a, w, out_dtype=dt, tile=256, **kw)
            r4 = ops.gemm_bf16(a, w, out_dtype=dt, tile=257, **kw)
            assert torch.equal(r8, r4), (K, sorted(kw))
            # the two-barrier deep-prefetch form (272) walks its tiles as one K-tile stream too (round 4; K >= 128: the
            # look-ahead reaches two K-tiles ahead) -- except with the generic epilogue, which stays one workgroup per tile
            r4d = ops.gemm_bf16(a, w, out_dtype=dt, tile=272, **kw)
            assert torch.equal(r8, r4d), (K, sorted(kw), "deep")
        # fp16 operands, the decoder's forms: SiLU * up with the fused row scale stays covered by the model-level tests; here
        # the in-place 16-bit residual epilogue through the persistent deep form against the one-barrier form
        a16, w16 = a.float().half(), w.float().half()
        r_a = ops.gemm_bf16(a16, w16, out_dtype=torch.float16, tile=257, silu_mul=True)
        r_b = ops.gemm_bf16(a16, w16, out_dtype=torch.float16, tile=272, silu_mul=True)
        assert torch.equal(r_a, r_b), (K, "fp16 silu deep")
        ref = a.float() @ w.float().T
        assert _rel(ops.gemm_bf16(a, w, out_dtype=torch.float32, tile=257), ref) < 2e-6


def test_gemm_timing_experiment_codes_are_refused(gpu):
    """Tile codes 261-267 are timing-only elimination experiments that compute wrong results: refused by the C ABI
    unless TCAVT_GEMM_TIMING_EXPERIMENTS is set in the environment."""
    import os

    from tcavt_amd import capi, ops

    if os.environ.get("TCAVT_GEMM_TIMING_EXPERIMENTS"):
        pytest.skip("experiments explicitly enabled in this environment")
    dev = gpu["device"]
    a = torch.zeros(256, 128, dtype=torch.bfloat16, device=dev)
    w = torch.zeros(256, 128, dtype=torch.bfloat16, device=dev)
    for code in (261, 264, 267):
        with pytest.raises(capi.TcavtError):
            ops.gemm_bf16(a, w, out_dtype=torch.bfloat16, silu_mul=True, tile=code)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [1, 7, 16, 20, 32])
def test_skinny_gemm_matches_tile_kernels(gpu, M, dt):
    """The skinny form tcavt_gemm_bf16 selects for M <= 32 rows (decode step of text generation: weights streamed once,
    eight waves split K) against the tiled kernels (forced tile) on every epilogue the decode step uses: fused-RMSNorm
    row scale + RoPE at per-row positions + LoRA second K source; row scale + SiLU*up; residual + 16-bit copy + partial
    sums of squares (one per 16 columns here, one per 64 there: the row sums must agree); plain fp32 (lm_head)."""
    import ctypes

    from tcavt_amd import capi, ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(100 + M)
    K, H = 512, 256
    x = (torch.randn(M, K, generator=g)).to(dt).to(dev)
    part = (torch.rand(M, 8, generator=g) * 40 + 10).to(dev)

    def run(tile, N, epi, w, out, **kw):
        a = capi.GemmArgs()
        a.A, a.lda, a.W, a.ldw, a.C, a.ldc = x.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), out.stride(0)
        a.M, a.N, a.K, a.tile, a.epilogue = M, N, K, tile, epi
        a.in_dtype, a.out_dtype = ops._DT[dt], ops._DT[out.dtype]
        for k_, v_ in kw.items():
            setattr(a, k_, v_.data_ptr() if torch.is_tensor(v_) else v_)
        capi.check(capi.lib().tcavt_gemm_bf16(ctypes.byref(a), capi.stream_ptr()), "gemm")
        return out

    rs = dict(rowscale_part=part, rowscale_npart=8, rowscale_h=K, rowscale_eps=1e-5)
    # q|k|v: RoPE at per-row positions, LoRA second source, row scale
    N = 384
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt).to(dev)
    t2 = torch.randn(M, 64, generator=g).to(dt).to(dev)
    w2 = (torch.randn(N, 64, generator=g) * 0.05).to(dt).to(dev)
    cos, sin = torch.rand(50, 32, generator=g).to(dev), torch.rand(50, 32, generator=g).to(dev)
    pos = torch.randint(0, 50, (M,), generator=g).to(torch.int32).to(dev)
    kw = dict(A2=t2, lda2=64, W2=w2, ldw2=64, K2=64, rope_cos=cos, rope_sin=sin, rope_L=50, rope_cols=320, rope_pos=pos, **rs)
    a_ = run(0, N, capi.EPI_ROPE | capi.EPI_ROWSCALE, w, torch.empty(M, N, dtype=dt, device=dev), **kw)
    b_ = run(128, N, capi.EPI_ROPE | capi.EPI_ROWSCALE, w, torch.empty(M, N, dtype=dt, device=dev), **kw)
    assert _rel(a_.float(), b_.float()) < (2e-3 if dt == torch.float16 else 8e-3)
    # gate|up: SiLU * up with row scale
    N = 512
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt).to(dev)
    a_ = run(0, N, capi.EPI_SILU_MUL | capi.EPI_ROWSCALE, w, torch.empty(M, N // 2, dtype=dt, device=dev), **rs)
    b_ = run(128, N, capi.EPI_SILU_MUL | capi.EPI_ROWSCALE, w, torch.empty(M, N // 2, dtype=dt, device=dev), **rs)
    assert _rel(a_.float(), b_.float()) < (2e-3 if dt == torch.float16 else 8e-3)
    # o / down: residual + 16-bit copy + partial sums of squares
    N = H
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    outs = []
    for tile, npart in ((0, ops.norm_npart(M, N, K)), (128, N // 64)):
        assert npart == (N // 16 if tile == 0 else N // 64)
        h16 = torch.zeros(M, N, dtype=dt, device=dev)
        pt = torch.zeros(M, npart, device=dev)
        c = run(tile, N, capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, w, torch.empty(M, N, device=dev), residual=res, ldr=N,
                norm_h16=h16, norm_part=pt)
        outs.append((c, h16, pt))
    (c0, h0, p0), (c1, h1, p1) = outs
    assert _rel(c0, c1) < 1e-5 and _rel(h0.float(), h1.float()) < 5e-3
    assert _rel(p0.sum(1), c0.pow(2).sum(1)) < 1e-5 and _rel(p1.sum(1), p0.sum(1)) < 1e-5
    assert _rel(c0, x.float() @ w.float().T + res) < 1e-5
    # lm_head: plain fp32 output, N not a multiple of the big tiles
    N = 1008
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt).to(dev)
    a_ = run(0, N, 0, w, torch.empty(M, N, device=dev))
    assert _rel(a_, x.float() @ w.float().T) < 1e-5


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [1, 8, 17, 32])
def test_skinny_gemm_fragment_major_weights_bit_equal(gpu, M, dt):
    """tcavt_pack_weight16 + tcavt_gemm_args.w_layout = W_FRAG16: the skinny form reads a fragment-major copy of W (1 KiB of
    consecutive bytes per wave instruction).  Same fragments into the same MFMAs in the same order: every epilogue form of the
    decode step must return BIT-identical results; the pack itself is checked against its definition; a tiled shape is refused."""
    import ctypes

    from tcavt_amd import capi, ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(900 + M)
    K, H = 1024, 512
    x = (torch.randn(M, K, generator=g)).to(dt).to(dev)
    part = (torch.rand(M, 8, generator=g) * 40 + 10).to(dev)

    def run(w, wl, N, epi, out, xin=None, **kw):
        a = capi.GemmArgs()
        a.A, a.lda, a.W, a.ldw = (x if xin is None else xin).data_ptr(), K, w.data_ptr(), K
        a.C, a.ldc = (out.data_ptr() if out is not None else None), kw.pop("ldc", N)
        a.M, a.N, a.K, a.tile, a.epilogue, a.w_layout = M, N, K, 0, epi, wl
        a.in_dtype, a.out_dtype = ops._DT[dt], ops._DT[out.dtype] if out is not None else capi.F32
        for k_, v_ in kw.items():
            setattr(a, k_, v_.data_ptr() if torch.is_tensor(v_) else v_)
        capi.check(capi.lib().tcavt_gemm_bf16(ctypes.byref(a), capi.stream_ptr()), "gemm")
        return out

    # the pack against its definition: chunk (b, j), lane l = 16 q + r <- W[16 b + r][32 j + 8 q : + 8]
    N = 384
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt).to(dev)
    wp = ops.pack_weight16(w)
    ref = w.view(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)
    assert torch.equal(wp.view(torch.int16), ref.view(torch.int16))
    rs = dict(rowscale_part=part, rowscale_npart=8, rowscale_h=K, rowscale_eps=1e-5)
    # q|k|v: RoPE + LoRA second source + row scale
    t2 = torch.randn(M, 64, generator=g).to(dt).to(dev)
    w2 = (torch.randn(N, 64, generator=g) * 0.05).to(dt).to(dev)
    cos, sin = torch.rand(50, 32, generator=g).to(dev), torch.rand(50, 32, generator=g).to(dev)
    pos = torch.randint(0, 50, (M,), generator=g).to(torch.int32).to(dev)
    kw = dict(A2=t2, lda2=64, W2=w2, ldw2=64, K2=64, rope_cos=cos, rope_sin=sin, rope_L=50, rope_cols=320, rope_pos=pos, **rs)
    a_ = run(w, 0, N, capi.EPI_ROPE | capi.EPI_ROWSCALE, torch.empty(M, N, dtype=dt, device=dev), **kw)
    b_ = run(wp, capi.W_FRAG16, N, capi.EPI_ROPE | capi.EPI_ROWSCALE, torch.empty(M, N, dtype=dt, device=dev), **kw)
    assert torch.equal(a_, b_) and a_.float().abs().max() > 0
    # gate|up: SiLU * up with row scale
    N = 512
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt).to(dev)
    wp = ops.pack_weight16(w)
    a_ = run(w, 0, N, capi.EPI_SILU_MUL | capi.EPI_ROWSCALE, torch.empty(M, N // 2, dtype=dt, device=dev), ldc=N // 2, **rs)
    b_ = run(wp, capi.W_FRAG16, N, capi.EPI_SILU_MUL | capi.EPI_ROWSCALE, torch.empty(M, N // 2, dtype=dt, device=dev), ldc=N // 2, **rs)
    assert torch.equal(a_, b_) and a_.float().abs().max() > 0
    # o / down: in-place 16-bit residual stream (fp16) or fp32 stream + 16-bit copy, partial sums of squares
    N = H
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt).to(dev)
    wp = ops.pack_weight16(w)
    res = torch.randn(M, N, generator=g).to(dev)
    outs = []
    for wt, wl in ((w, 0), (wp, capi.W_FRAG16)):
        npart = ops.norm_npart(M, N, K)
        h16 = res.to(dt)
        pt = torch.zeros(M, npart, device=dev)
        if dt == torch.float16:
            run(wt, wl, N, capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, None, norm_h16=h16, norm_part=pt)
            outs.append((h16, pt))
        else:
            c = run(wt, wl, N, capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, torch.empty(M, N, device=dev), residual=res, ldr=N, norm_h16=h16,
                    norm_part=pt)
            outs.append((c, h16, pt))
    for u_, v_ in zip(*outs):
        assert torch.equal(u_, v_)
    assert not torch.equal(outs[0][0].float(), res.to(dt).float())
    # lm_head: plain fp32 output (both column-block forms: N < 8192 and N >= 8192 with M > 16)
    for N in (1008, 8192):
        w = (torch.randn(N, K, generator=g) * 0.05).to(dt).to(dev)
        wp = ops.pack_weight16(w)
        a_ = run(w, 0, N, 0, torch.empty(M, N, device=dev))
        b_ = run(wp, capi.W_FRAG16, N, 0, torch.empty(M, N, device=dev))
        assert torch.equal(a_, b_) and _rel(a_, x.float() @ w.float().T) < 1e-5
    # ---- fragment-major ACTIVATIONS as well (tcavt_gemm_args.act_layout): A read, SiLU output / in-place stream written in the
    # operand order of the next skinny GEMM; the layout is the weight pack's with tokens as rows (padded to whole 16-row blocks)
    Mr = 16 if M <= 16 else 32
    AF, AO = capi.ACT_A_FRAG16, capi.ACT_A_FRAG16 | capi.ACT_OUT_FRAG16

    def to_frag(t):  # [M, C] -> fragment-major [Mr * C]
        pad = torch.zeros(Mr, t.shape[1], dtype=t.dtype, device=dev)
        pad[:M] = t
        return ops.pack_weight16(pad)

    def from_frag(f, C):  # inverse, first M rows
        return f.view(Mr // 16, C // 32, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(Mr, C)[:M]

    _frag_activation_forms(M, dt, dev, g, run, x, rs, kw, K, H, to_frag, from_frag, AF, AO, Mr)
    if M <= 8:  # ONE block of 8 tokens (TCAVT_ACT_BLOCK8): element (m, f) at (f >> 5) * 256 + ((f >> 3) & 3) * 64 + m * 8 + (f & 7)
        def to_frag8(t):
            pad = torch.zeros(8, t.shape[1], dtype=t.dtype, device=dev)
            pad[:M] = t
            return pad.view(8, t.shape[1] // 32, 4, 8).permute(1, 2, 0, 3).contiguous().view(-1)

        def from_frag8(f, C):
            return f.view(C // 32, 4, 8, 8).permute(2, 0, 1, 3).reshape(8, C)[:M]

        _frag_activation_forms(M, dt, dev, g, run, x, rs, kw, K, H, to_frag8, from_frag8, AF | capi.ACT_BLOCK8, AO | capi.ACT_BLOCK8, 8)
    # refused where no skinny form runs (M > 32)
    xl = torch.zeros(64, K, dtype=dt, device=dev)
    a = capi.GemmArgs()
    o = torch.empty(64, N, device=dev)
    a.A, a.lda, a.W, a.ldw, a.C, a.ldc, a.M, a.N, a.K = xl.data_ptr(), K, wp.data_ptr(), K, o.data_ptr(), N, 64, N, K
    a.in_dtype, a.out_dtype, a.w_layout = ops._DT[dt], capi.F32, capi.W_FRAG16
    assert capi.lib().tcavt_gemm_bf16(ctypes.byref(a), capi.stream_ptr()) != 0


def _frag_activation_forms(M, dt, dev, g, run, x, rs, kw, K, H, to_frag, from_frag, AF, AO, Mr):
    """Every epilogue form of the decode step with fragment-major activations (A read / 16-bit result written in the next skinny
    GEMM's operand order; AF / AO = the act_layout flags of an A-only / A-and-output call) against the row-major call:
    bit-identical."""
    from tcavt_amd import capi, ops

    xf = to_frag(x)
    N = 512
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt).to(dev)
    wp = ops.pack_weight16(w)
    a_ = run(w, 0, N, capi.EPI_SILU_MUL | capi.EPI_ROWSCALE, torch.empty(M, N // 2, dtype=dt, device=dev), ldc=N // 2, **rs)
    of = torch.zeros(Mr * (N // 2), dtype=dt, device=dev)
    run(wp, capi.W_FRAG16, N, capi.EPI_SILU_MUL | capi.EPI_ROWSCALE, of, xin=xf, ldc=N // 2, act_layout=AO, **rs)
    assert torch.equal(from_frag(of, N // 2), a_)
    b_ = run(wp, capi.W_FRAG16, N, capi.EPI_SILU_MUL | capi.EPI_ROWSCALE, torch.empty(M, N // 2, dtype=dt, device=dev), xin=xf,
             ldc=N // 2, act_layout=AF, **rs)
    assert torch.equal(b_, a_)
    # q|k|v (row-major output) and lm_head from fragment-major A
    Nq_ = 384
    wq = (torch.randn(Nq_, K, generator=g) * 0.05).to(dt).to(dev)
    qa = run(wq, 0, Nq_, capi.EPI_ROPE | capi.EPI_ROWSCALE, torch.empty(M, Nq_, dtype=dt, device=dev), **kw)
    la = run(wq, 0, Nq_, 0, torch.empty(M, Nq_, device=dev))
    qb = run(ops.pack_weight16(wq), capi.W_FRAG16, Nq_, capi.EPI_ROPE | capi.EPI_ROWSCALE, torch.empty(M, Nq_, dtype=dt, device=dev),
             xin=xf, act_layout=AF, **kw)
    lb = run(wq, 0, Nq_, 0, torch.empty(M, Nq_, device=dev), xin=xf, act_layout=AF)
    assert torch.equal(qa, qb) and torch.equal(la, lb)
    # o / down: the in-place 16-bit stream in fragment-major order
    N = H
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt).to(dev)
    res16 = torch.randn(M, N, generator=g).to(dev).to(dt)
    npart = ops.norm_npart(M, N, K)
    h_a, p_a = res16.clone(), torch.zeros(M, npart, device=dev)
    run(w, 0, N, capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, None, norm_h16=h_a, norm_part=p_a)
    h_b, p_b = to_frag(res16), torch.zeros(M, npart, device=dev)
    run(ops.pack_weight16(w), capi.W_FRAG16, N, capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, None, xin=xf, norm_h16=h_b, norm_part=p_b,
        act_layout=AO)
    assert torch.equal(from_frag(h_b, N), h_a) and torch.equal(p_a, p_b) and not torch.equal(h_a, res16)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [1, 8, 32])
def test_skinny_gemm_split_k_across_workgroups(gpu, M, dt):
    """tcavt_gemm_args.splitk_ws: with a workspace the skinny form splits K over several workgroups per column block (fp32
    slabs + ticket, the last arriver adds the slices in order and runs the epilogue).  At the decode step's own shapes
    (Llama-3.2-1B: q|k|v 3072 x 2048 RoPE + LoRA + row scale, o 2048 x 2048 and down 2048 x 8192 residual + 16-bit copy +
    sums of squares, gate|up 16384 x 2048 SiLU) the result must equal the one-workgroup form to summation-order noise, be
    bit-identical from launch to launch, leave the tickets at zero, and the LoRA source must be added exactly once."""
    import ctypes

    from tcavt_amd import capi, ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(4100 + M)
    ws = torch.zeros(9 << 20, dtype=torch.uint8, device=dev)
    ws[16 << 10:] = 0x7F  # (slabs start as garbage: every word read must have been written by this launch)

    def run(x, K, N, epi, w, out, split, **kw):
        a = capi.GemmArgs()
        a.A, a.lda, a.W, a.ldw, a.C, a.ldc = x.data_ptr(), K, w.data_ptr(), K, None if out is None else out.data_ptr(), N if out is None else out.stride(0)
        a.M, a.N, a.K, a.tile, a.epilogue = M, N, K, 0, epi
        a.in_dtype, a.out_dtype = ops._DT[dt], capi.F32 if out is None else ops._DT[out.dtype]
        if split:
            a.splitk_ws, a.splitk_ws_bytes = ws.data_ptr(), ws.numel()
        for k_, v_ in kw.items():
            setattr(a, k_, v_.data_ptr() if torch.is_tensor(v_) else v_)
        capi.check(capi.lib().tcavt_gemm_bf16(ctypes.byref(a), capi.stream_ptr()), "gemm")

    def tickets_clear():
        return int(ws[: 16 << 10].view(torch.int32).abs().sum()) == 0

    tol = 2e-3 if dt == torch.float16 else 8e-3
    K = 2048
    x = torch.randn(M, K, generator=g).to(dt).to(dev)
    part = (torch.rand(M, 128, generator=g) * 40 + 10).to(dev)
    rs = dict(rowscale_part=part, rowscale_npart=128, rowscale_h=K, rowscale_eps=1e-5)
    # q|k|v
    N = 3072
    w = (torch.randn(N, K, generator=g) * 0.03).to(dt).to(dev)
    t2 = torch.randn(M, 64, generator=g).to(dt).to(dev)
    w2 = (torch.randn(N, 64, generator=g) * 0.2).to(dt).to(dev)
    cos, sin = torch.rand(50, 32, generator=g).to(dev), torch.rand(50, 32, generator=g).to(dev)
    pos = torch.randint(0, 50, (M,), generator=g).to(torch.int32).to(dev)
    kw = dict(A2=t2, lda2=64, W2=w2, ldw2=64, K2=64, rope_cos=cos, rope_sin=sin, rope_L=50, rope_cols=2560, rope_pos=pos, **rs)
    o = [torch.empty(M, N, dtype=dt, device=dev) for _ in range(3)]
    run(x, K, N, capi.EPI_ROPE | capi.EPI_ROWSCALE, w, o[0], False, **kw)
    run(x, K, N, capi.EPI_ROPE | capi.EPI_ROWSCALE, w, o[1], True, **kw)
    run(x, K, N, capi.EPI_ROPE | capi.EPI_ROWSCALE, w, o[2], True, **kw)
    assert _rel(o[1].float(), o[0].float()) < tol and torch.equal(o[1], o[2]) and tickets_clear()
    # gate|up
    N = 16384
    w = (torch.randn(N, K, generator=g) * 0.03).to(dt).to(dev)
    o = [torch.empty(M, N // 2, dtype=dt, device=dev) for _ in range(3)]
    for i in range(3):
        run(x, K, N, capi.EPI_SILU_MUL | capi.EPI_ROWSCALE, w, o[i], i > 0, **rs)
    assert _rel(o[1].float(), o[0].float()) < tol and torch.equal(o[1], o[2]) and tickets_clear()
    # o (K = 2048) and down (K = 8192): fp32 stream and 16-bit stream
    changed = False
    for K in (2048, 8192):
        N = 2048
        x = torch.randn(M, K, generator=g).to(dt).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.03).to(dt).to(dev)
        res = torch.randn(M, N, generator=g).to(dev)
        outs = []
        for i in range(3):
            h16 = torch.zeros(M, N, dtype=dt, device=dev)
            pt = torch.zeros(M, N // 16, device=dev)
            c = torch.empty(M, N, device=dev)
            run(x, K, N, capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, w, c, i > 0, residual=res, ldr=N, norm_h16=h16, norm_part=pt)
            outs.append((c, h16, pt))
        assert _rel(outs[1][0], outs[0][0]) < 1e-5 and _rel(outs[1][2].sum(1), outs[0][2].sum(1)) < 1e-5
        assert _rel(outs[1][0], x.float() @ w.float().T + res) < 1e-5
        assert all(torch.equal(a_, b_) for a_, b_ in zip(outs[1], outs[2])) and tickets_clear()
        changed |= not torch.equal(outs[1][0], outs[0][0])
        outs = []
        for i in range(3):
            h16 = res.to(dt)
            pt = torch.zeros(M, N // 16, device=dev)
            run(x, K, N, capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, w, None, i > 0, norm_h16=h16, norm_part=pt)  # (in place: h16 += x W^T)
            outs.append((h16, pt))
        assert _rel(outs[1][0].float(), outs[0][0].float()) < tol and _rel(outs[1][1].sum(1), outs[0][1].sum(1)) < tol
        assert all(torch.equal(a_, b_) for a_, b_ in zip(outs[1], outs[2])) and tickets_clear()
    assert changed  # (a different summation order: the split really ran)


@pytest.mark.parametrize("stream16", [True, False])
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [1, 8, 13, 20, 32])
def test_skinny_gemm_lora_down_from_partial_sums(gpu, M, dt, stream16):
    """tcavt_gemm_args.lora_part (decode step): the down-projection GEMM's workgroups leave the partial dot products of their
    16 columns of the rounded residual stream with the next layer's adapter rows, and the q|k|v GEMM adds them up in place of
    the separate LoRA down-projection launch.  Against the two-launch form (t = scale * h16 . a_cat^T as a GEMM of its own,
    then A2 = t) at the decode step's shapes: equal up to one 16-bit rounding of t (its summation order differs), and the
    adapter term must be a visible part of the output."""
    import ctypes

    from tcavt_amd import capi, ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(5200 + M)
    H, I, NQKV, r, LV = 2048, 8192, 3072, 8, 16
    x = torch.randn(M, I, generator=g).to(dt).to(dev)
    w_d = (torch.randn(H, I, generator=g) * 0.02).to(dt).to(dev)
    res = torch.randn(M, H, generator=g).to(dev)
    a_cat = torch.zeros(64, H)
    a_cat[:r] = torch.randn(r, H, generator=g) * 0.05
    a_cat[LV:LV + r] = torch.randn(r, H, generator=g) * 0.05
    a_cat = a_cat.to(dt).to(dev)
    b_ext = torch.zeros(NQKV, 64)
    b_ext[:2048, :r] = torch.randn(2048, r, generator=g) * 0.3
    b_ext[2560:, LV:LV + r] = torch.randn(512, r, generator=g) * 0.3
    b_ext = b_ext.to(dt).to(dev)
    w_qkv = (torch.randn(NQKV, H, generator=g) * 0.03).to(dt).to(dev)
    cos, sin = torch.rand(50, 32, generator=g).to(dev), torch.rand(50, 32, generator=g).to(dev)
    pos = torch.randint(0, 50, (M,), generator=g).to(torch.int32).to(dev)
    scale = 4.0

    def gemm(A, K, W, N, out, epi, **kw):
        a = capi.GemmArgs()
        a.A, a.lda, a.W, a.ldw = A.data_ptr(), K, W.data_ptr(), K
        a.C, a.ldc = (None, N) if out is None else (out.data_ptr(), out.stride(0))
        a.M, a.N, a.K, a.tile, a.epilogue = M, N, K, 0, epi
        a.in_dtype, a.out_dtype = ops._DT[dt], capi.F32 if out is None else ops._DT[out.dtype]
        for k_, v_ in kw.items():
            setattr(a, k_, v_.data_ptr() if torch.is_tensor(v_) else v_)
        capi.check(capi.lib().tcavt_gemm_bf16(ctypes.byref(a), capi.stream_ptr()), "gemm")

    def down(lp):
        h16 = res.to(dt) if stream16 else torch.zeros(M, H, dtype=dt, device=dev)
        part = torch.zeros(M, H // 16, device=dev)
        kw = dict(norm_h16=h16, norm_part=part)
        c = None
        if not stream16:
            c = torch.empty(M, H, device=dev)
            kw.update(residual=res, ldr=H)
        if lp is not None:
            kw.update(lora_part=lp, lora_part_a=a_cat, lora_part_lda=H)
        gemm(x, I, w_d, H, c, capi.EPI_RESIDUAL | capi.EPI_NORM_OUT, **kw)
        return h16, part

    rope = dict(rope_cos=cos, rope_sin=sin, rope_L=50, rope_cols=2560, rope_pos=pos)
    # two launches
    h16, part = down(None)
    t = torch.zeros(M, 64, dtype=dt, device=dev)
    gemm(h16, H, a_cat, 64, t, 0, acc_scale=scale)
    rs = dict(rowscale_part=part, rowscale_npart=H // 16, rowscale_h=H, rowscale_eps=1e-5)
    want = torch.empty(M, NQKV, dtype=dt, device=dev)
    gemm(h16, H, w_qkv, NQKV, want, capi.EPI_ROPE | capi.EPI_ROWSCALE, A2=t, lda2=64, W2=b_ext, ldw2=64, K2=64, **rope, **rs)
    plain = torch.empty(M, NQKV, dtype=dt, device=dev)
    gemm(h16, H, w_qkv, NQKV, plain, capi.EPI_ROPE | capi.EPI_ROWSCALE, **rope, **rs)
    # partial sums
    lp = torch.full((H // 16, M, 16), float("nan"), device=dev)
    h16b, partb = down(lp)
    assert torch.equal(h16b, h16) and torch.equal(partb, part)
    t_ref = (h16.float() @ a_cat.float().T)[:, list(range(r)) + list(range(LV, LV + r))]
    assert _rel(lp.sum(0), t_ref) < 1e-5
    got = [torch.empty(M, NQKV, dtype=dt, device=dev) for _ in range(2)]
    for o in got:
        gemm(h16, H, w_qkv, NQKV, o, capi.EPI_ROPE | capi.EPI_ROWSCALE, W2=b_ext, ldw2=64, lora_part=lp, lora_part_np=H // 16,
             lora_part_scale=scale, **rope, **rs)
    assert torch.equal(got[0], got[1])
    ulp = 2.0 ** (-10 if dt == torch.float16 else -7)
    assert _rel(want.float(), plain.float()) > 0.05                 # the adapters matter here
    assert _rel(got[0].float(), want.float()) < 2 * ulp             # ... and arrive the same way (t within one rounding)
    assert torch.equal(got[0][:, 2048:2560], want[:, 2048:2560])    # k rows of b_ext are zero: untouched
    # refused outside the decode step's form
    with pytest.raises(capi.TcavtError):
        gemm(h16, H, w_qkv, NQKV, got[0], capi.EPI_ROPE | capi.EPI_ROWSCALE, W2=b_ext, ldw2=64, lora_part=lp, lora_part_np=0,
             lora_part_scale=scale, **rope, **rs)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,tile", [(300, 128), (300, 256), (512, 257), (512, 272), (512, 0), (20, 0)])
def test_gemm_norm_out_fp32_and_16bit_stream(gpu, M, tile, dt):
    """TCAVT_EPI_NORM_OUT on every kernel form that serves it, in both residual-stream modes:
    fp32 stream: C = residual + A W^T, 16-bit copy, partial sums of squares of the fp32 values;
    16-bit stream (C == NULL): norm_h16 <- round(norm_h16 + A W^T) in place, partial sums of the ROUNDED values."""
    import ctypes

    from tcavt_amd import capi, ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(7 * M + tile)
    K, N = 512, 512
    x = torch.randn(M, K, generator=g).to(dt).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    npart = ops.norm_npart(M, N, K) if tile == 0 else N // 64
    gw = N // npart

    def run(C, residual, h16, pt, epi):
        a = capi.GemmArgs()
        a.A, a.lda, a.W, a.ldw, a.ldc = x.data_ptr(), K, w.data_ptr(), K, N
        a.C = None if C is None else C.data_ptr()
        if residual is not None:
            a.residual, a.ldr = residual.data_ptr(), N
        a.M, a.N, a.K, a.tile, a.epilogue = M, N, K, tile, epi
        a.in_dtype, a.out_dtype = ops._DT[dt], capi.F32
        a.norm_h16, a.norm_part = h16.data_ptr(), pt.data_ptr()
        return capi.lib().tcavt_gemm_bf16(ctypes.byref(a), capi.stream_ptr())

    prod = x.float() @ w.float().T
    for with_res in (True, False):
        epi = capi.EPI_NORM_OUT | (capi.EPI_RESIDUAL if with_res else 0)
        # fp32 stream
        C, h16, pt = torch.empty(M, N, device=dev), torch.zeros(M, N, dtype=dt, device=dev), torch.zeros(M, npart, device=dev)
        capi.check(run(C, res if with_res else None, h16, pt, epi), "gemm")
        want = prod + (res if with_res else 0)
        assert _rel(C, want) < 1e-5
        assert torch.equal(h16, C.to(dt))
        assert _rel(pt, C.view(M, npart, gw).pow(2).sum(-1)) < 1e-5
        # 16-bit stream, in place
        s0 = res.to(dt)
        s16, pt2 = s0.clone(), torch.zeros(M, npart, device=dev)
        capi.check(run(None, None, s16, pt2, epi), "gemm")
        want16 = prod + (s0.float() if with_res else 0)
        # (the accumulation order differs from torch's: values at a rounding boundary may land one 16-bit ulp apart)
        ulp = 2.0 ** (-10 if dt == torch.float16 else -7)
        assert ((s16.float() - want16).abs() <= ulp * want16.abs().clamp_min(2.0 ** -14) * 1.01 + 2e-5).all()  # (+ fp32 accumulation-order noise)
        assert _rel(s16.float(), want16) < (5e-4 if dt == torch.float16 else 4e-3)
        assert _rel(pt2, s16.float().view(M, npart, gw).pow(2).sum(-1)) < 1e-5  # sums of what is stored
    # C == NULL with a residual pointer is refused (the stream is norm_h16 itself)
    assert run(None, res, s16, pt2, capi.EPI_NORM_OUT | capi.EPI_RESIDUAL) != 0


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(1024, 2048, 4096), (1024, 2048, 8192), (2048, 2048, 8192), (900, 512, 16384)])
def test_gemm_norm_out_two_launch_split_k(gpu, M, N, K, dt):
    """The in-place 16-bit residual form on a grid that cannot fill the chip (config 4's small batches): with a workspace the
    product runs as S partial products (batched launch into fp32 slabs) + one reduce-and-epilogue kernel.  Same contract as the
    one-launch form (stream updated in place or out of place, partial sums of the ROUNDED values per 64 columns, norm_scale),
    bit-reproducible run to run; the K order of the sum differs, so the two forms agree to rounding, not bit for bit."""
    import ctypes

    from tcavt_amd import capi, ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) * 0.5).to(dt).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.03).to(dt).to(dev)
    s0 = (torch.randn(M, N, generator=g) * 2.0).to(dt).to(dev)
    wsp = torch.zeros((16 << 10) + 8 * 1024 * 2048 * 4, dtype=torch.uint8, device=dev)
    npart = N // 64

    def run(ws, nscale=0.0, out_of_place=False):
        a = capi.GemmArgs()
        a.A, a.lda, a.W, a.ldw, a.ldc, a.C = x.data_ptr(), K, w.data_ptr(), K, N, None
        a.M, a.N, a.K, a.tile, a.epilogue = M, N, K, 0, capi.EPI_NORM_OUT | capi.EPI_RESIDUAL
        a.in_dtype, a.out_dtype = ops._DT[dt], capi.F32
        src = s0.clone() if nscale == 0.0 else (s0.float() * nscale).to(dt)
        dst = torch.zeros_like(src) if out_of_place else src
        pt = torch.zeros(M, npart, device=dev)
        a.norm_h16, a.norm_part, a.norm_scale = dst.data_ptr(), pt.data_ptr(), nscale
        if out_of_place:
            a.norm_res16 = src.data_ptr()
        if ws is not None:
            a.splitk_ws, a.splitk_ws_bytes = ws.data_ptr(), ws.numel()
        capi.check(capi.lib().tcavt_gemm_bf16(ctypes.byref(a), capi.stream_ptr()), "gemm")
        torch.cuda.synchronize()
        return dst, pt

    one, p_one = run(None)
    two, p_two = run(wsp)
    two_b, p_two_b = run(wsp)
    assert torch.equal(two, two_b) and torch.equal(p_two, p_two_b)  # slice-order sum: reproducible
    want = x.float() @ w.float().T + s0.float()
    ulp = 2.0 ** (-10 if dt == torch.float16 else -7)
    assert ((two.float() - want).abs() <= ulp * want.abs().clamp_min(2.0 ** -14) * 1.01 + 1e-4).all()
    assert _rel(two.float(), one.float()) < (3e-4 if dt == torch.float16 else 3e-3)
    assert _rel(p_two, two.float().view(M, npart, 64).pow(2).sum(-1)) < 1e-5  # sums of what is stored
    assert (two != one).float().mean().item() < 0.2  # (most elements are even bit-equal: only rounding-boundary cases move)
    # out of place (the LoRA-trainable variant's tape) and at a stream scale
    oop, p_oop = run(wsp, out_of_place=True)
    assert torch.equal(oop, two) and torch.equal(p_oop, p_two)
    sc, p_sc = run(wsp, nscale=2.0 ** -3)
    want_s = (x.float() @ w.float().T) * 2.0 ** -3 + (s0.float() * 2.0 ** -3).to(dt).float()
    assert _rel(sc.float(), want_s) < (5e-4 if dt == torch.float16 else 4e-3)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,H", [(37, 256), (8, 2048), (5, 8192), (3, 768), (9, 4352)])
def test_rmsnorm16(gpu, M, H, dt):
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(M + H)
    x = (torch.randn(M, H, generator=g) * 3).to(dt).to(dev)
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(dev)
    o32 = torch.empty(M, H, device=dev)
    o16 = torch.empty(M, H, dtype=dt, device=dev)
    ops.rmsnorm16(x, gamma, 1e-5, out16=o16, out_f32=o32)
    xf = x.float()
    want = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5) * gamma
    assert _rel(o32, want) < 1e-6
    assert torch.equal(o16, o32.to(dt))


def test_embed_fuse_16bit_stream(gpu):
    """tcavt_embed_fuse with h == NULL: the 16-bit stream equals the rounded fp32 embedding rows, the partial sum (slot 0)
    is that of the ROUNDED values; tcavt_rownorm_prep(rounded_sums) likewise."""
    from tcavt_amd import ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(3)
    B, Nq, Lt, H, V = 2, 4, 9, 256, 50
    table = torch.randn(V, H, generator=g).to(torch.float16).to(dev)
    ids = torch.randint(0, V, (B, Lt), generator=g).to(dev)
    img = torch.randn(B * Nq, H, generator=g).to(dev)
    vis, txt = torch.randn(H, generator=g).to(dev), torch.randn(H, generator=g).to(dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    rows = B * (Nq + Lt)
    h = torch.empty(rows, H, device=dev)
    h16a, pa = torch.empty(rows, H, dtype=torch.float16, device=dev), torch.empty(rows, 4, device=dev)
    ops.embed_fuse(table, ids, img, vis, txt, h, flag, h16=h16a, part=pa, npart=4)
    h16b, pb = torch.empty_like(h16a), torch.empty_like(pa)
    ops.embed_fuse(table, ids, img, vis, txt, None, flag, h16=h16b, part=pb, npart=4)
    assert torch.equal(h16a, h16b) and torch.equal(h16b, h.to(torch.float16))
    assert _rel(pa[:, 0], h.pow(2).sum(-1)) < 1e-6 and _rel(pb[:, 0], h16b.float().pow(2).sum(-1)) < 1e-6
    assert (pb[:, 1:] == 0).all()
    h16c, pc = torch.empty_like(h16a), torch.empty_like(pa)
    ops.rownorm_prep(h, h16c, pc, npart=4, rounded_sums=True)
    assert torch.equal(h16c, h16b) and _rel(pc[:, 0], pb[:, 0]) < 1e-6
    assert flag.item() == 0


@pytest.mark.parametrize("train", [False, True])
def test_tlayer_stack_matches_python_composition(gpu, train, monkeypatch):
    """tcavt_tlayer_stack_forward (Q-Former encoder + decoder stacks in 16-bit, lane-polygon encoder in fp32, as ONE C call each)
    and tcavt_cross_attn_forward (the head's absorbed cross-attention) against the per-launch Python composition of the same
    kernel-level entry points (TCAVT_PY_TLAYERS=1): bit-identical image tokens, polygon embeddings and decoded trajectories,
    eval arithmetic and train mode (same dropout sites)."""
    from tcavt_amd import model
    from tests.util import batch_tensors, load_case

    dev = gpu["device"]
    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    t = batch_tensors(fx)
    g = {k: v.to(dev) for k, v in t.items()}
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev)
    m.train(train)

    def run(py):
        if py:
            monkeypatch.setenv("TCAVT_PY_TLAYERS", "1")
        else:
            monkeypatch.delenv("TCAVT_PY_TLAYERS", raising=False)
        dc = model.DropoutCtx(1234) if train else None
        m.mllm.qformer.dctx = dc.sub(1) if dc else None
        m.lane_polygon_encoder.dctx = dc.sub(0) if dc else None
        with torch.no_grad():
            img = m.mllm._image_tokens(g["vision_emb"]).clone()
            emb = m.lane_polygon_encoder(g["lane_polygon"], g["lane_polygon_len"]).clone()
            # the whole model: also the head's cross-attention (tcavt_cross_attn_forward vs its per-launch composition)
            m._fwd_count = 0
            dec = m(g["traj_emb"], g["vision_emb"], None, g["lane_polygon"], g["lane_polygon_len"], input_ids=g["input_ids"],
                    attention_mask=g["attention_mask"]).clone()
        torch.cuda.synchronize()
        return img, emb, dec

    img_c, emb_c, dec_c = run(False)
    img_p, emb_p, dec_p = run(True)
    assert torch.isfinite(img_c).all() and torch.isfinite(emb_c).all() and torch.isfinite(dec_c).all()
    assert torch.equal(img_c, img_p) and torch.equal(emb_c, emb_p) and torch.equal(dec_c, dec_p)


def test_gemm_norm_out_16bit_stream_out_of_place(gpu):
    """tcavt_gemm_args.norm_res16: the in-place 16-bit residual epilogue reading the stream from one buffer and writing it to
    another (the LoRA-trainable variant's per-layer stream tape) gives the bits of the in-place form, and leaves the source
    untouched -- 4-wave kernel (16-byte accesses) and the small-tile kernel."""
    import ctypes

    from tcavt_amd import capi, ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(4)
    for M, N, K in ((512, 512, 256), (96, 128, 64)):
        a = (torch.randn(M, K, generator=g) * 0.3).to(torch.float16).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.05).to(torch.float16).to(dev)
        h0 = torch.randn(M, N, generator=g).to(torch.float16).to(dev)

        def run(src, dst, part):
            ga = capi.GemmArgs()
            ga.A, ga.lda, ga.W, ga.ldw, ga.C, ga.ldc = a.data_ptr(), K, w.data_ptr(), K, None, N
            ga.M, ga.N, ga.K = M, N, K
            ga.in_dtype, ga.out_dtype = capi.F16, capi.F32
            ga.epilogue = capi.EPI_RESIDUAL | capi.EPI_NORM_OUT
            ga.norm_h16, ga.norm_part = dst.data_ptr(), part.data_ptr()
            if src is not dst:
                ga.norm_res16 = src.data_ptr()
            capi.check(capi.lib().tcavt_gemm_bf16(ctypes.byref(ga), capi.stream_ptr()), "gemm")

        inpl, p0 = h0.clone(), torch.zeros(M, N // 64, device=dev)
        run(inpl, inpl, p0)
        src, dst, p1 = h0.clone(), torch.zeros_like(h0), torch.zeros(M, N // 64, device=dev)
        run(src, dst, p1)
        torch.cuda.synchronize()
        assert torch.equal(dst, inpl) and torch.equal(p1, p0) and torch.equal(src, h0)
        ref = h0.float() + a.float() @ w.float().T
        assert _rel(dst.float(), ref) < 1e-3
