"""-m gpu: pda_linear_split (csrc/gemm_split.hip) -- f32 GEMM on the bf16 matrix cores with every operand split into three
bf16 terms -- against an f64 product of the same f32 inputs.

The bar is f32 arithmetic, not bf16: the error relative to sum_k |x_k w_k| must stay at the level of an f32 fmaf chain
(the library's f32 GEMM and this repo's f32-MFMA kernel measure 6e-7 .. 9e-7 on these inputs; 2e-6 is asserted), three
orders of magnitude below a single-bf16 product (4e-3)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel_err(y, x, w, bias):
    ref = x.double() @ w.double().t() + (0 if bias is None else bias.double())
    scale = x.double().abs() @ w.double().abs().t() + 1e-30
    return ((y.double() - ref).abs() / scale).max().item()


@pytest.mark.parametrize("T,K,N", [(1, 32, 128), (127, 64, 128), (1000, 96, 256), (4099, 128, 384), (20000, 256, 512),
                                   (33000, 512, 512), (5000, 384, 1536), (2048, 512, 2048),
                                   # few tokens: the output chunks spread over blockIdx.y (2 chunks per workgroup; a ragged last group)
                                   (6600, 192, 1280), (16384, 128, 384), (4096, 256, 256)])
def test_split_gemm_has_f32_accuracy(T, K, N):
    from pdanet_amd import pointnet2_batch_cuda as ext
    g = torch.Generator("cuda").manual_seed(T + K + N)
    # wide dynamic range: magnitudes over ~6 decades, so the low split terms matter
    x = torch.randn(T, K, device="cuda", generator=g) * torch.exp(3 * torch.randn(T, K, device="cuda", generator=g))
    w = torch.randn(N, K, device="cuda", generator=g) * torch.exp(torch.randn(N, K, device="cuda", generator=g))
    bias = torch.randn(N, device="cuda", generator=g)
    wf = ext.linear_split_pack(w, N, K)
    y = torch.full((T, N), float("nan"), device="cuda")
    ext.linear_split(x, wf, bias, y, T, K, N)
    e = _rel_err(y, x, w, bias)
    e_lib = _rel_err(torch.nn.functional.linear(x, w, bias), x, w, bias)
    assert e < 2e-6 and e < 3 * e_lib + 1e-7, (e, e_lib)
    # the packed planes do not depend on the source orientation
    assert torch.equal(wf, ext.linear_split_pack(w.t().contiguous(), N, K, transposed_source=True))
    # no bias, fused ReLU
    y2 = torch.empty(T, N, device="cuda")
    ext.linear_split(x, wf, None, y2, T, K, N, relu=True)
    assert torch.equal(y2, torch.relu(y2)) and _rel_err(torch.where(y2 > 0, y2, (y - bias).clamp(max=0)), x, w, None) < 2e-6


def test_split_gemm_exact_on_integers_and_refuses_other_shapes():
    """Small integers are exact in bf16: the product must be exact (catches a wrong fragment / k order, which random
    data with a tolerance could hide for symmetric inputs)."""
    from pdanet_amd import pointnet2_batch_cuda as ext, _lib
    T, K, N = 333, 192, 256
    g = torch.Generator("cuda").manual_seed(7)
    x = torch.randint(-8, 9, (T, K), device="cuda", generator=g).float()
    w = torch.randint(-8, 9, (N, K), device="cuda", generator=g).float()
    y = torch.empty(T, N, device="cuda")
    ext.linear_split(x, ext.linear_split_pack(w, N, K), None, y, T, K, N)
    assert torch.equal(y, (x.double() @ w.double().t()).float())
    with pytest.raises(_lib.PdaError):
        ext.linear_split(x[:, :100].contiguous(), ext.linear_split_pack(w, N, K), None, y, T, 100, N)
    with pytest.raises(_lib.PdaError):
        ext.linear_split(x, ext.linear_split_pack(w, N, K), None, y, T, K, 200)


@pytest.mark.parametrize("T,K,N", [(1, 32, 1), (100, 64, 100), (4099, 32, 259), (5000, 768, 256), (12979, 256, 259),   # 128 x 128 tiles
                                   (33000, 512, 512), (64394, 1536, 512), (40000, 512, 1536), (70001, 256, 640)])   # 256 x 256 tiles
def test_tiled_split_gemm(T, K, N):
    """pda_gemm_split (gemm_split_kernel / gemm_split_wide_kernel): any K % 32 == 0, any N, ragged T; bias, fused ReLU and the
    accumulate form; the same f32-grade bar as above."""
    from pdanet_amd import pointnet2_batch_cuda as ext
    g = torch.Generator("cuda").manual_seed(T + K + N)
    x = torch.randn(T, K, device="cuda", generator=g) * torch.exp(2 * torch.randn(T, K, device="cuda", generator=g))
    w = torch.randn(N, K, device="cuda", generator=g)
    bias = torch.randn(N, device="cuda", generator=g)
    wf = ext.linear_split_pack(w, N, K)
    y = torch.full((T, N), float("nan"), device="cuda")
    ext.gemm_split(x, wf, bias, y, T, K, N)
    assert torch.isfinite(y).all()
    e, e_lib = _rel_err(y, x, w, bias), _rel_err(torch.nn.functional.linear(x, w, bias), x, w, bias)
    assert e < 4e-6 and e < 3 * e_lib + 1e-7, (e, e_lib)      # f32 accumulation over K up to 1536: the library's own error level
    # dX = dY W through the transposed pack, accumulated onto an existing tensor
    gy = torch.randn(T, N, device="cuda", generator=g)
    if N % 32 == 0:
        base = torch.randn(T, K, device="cuda", generator=g)
        acc = base.clone()
        ext.gemm_split(gy, ext.linear_split_pack(w, K, N, transposed_source=True), None, acc, T, N, K, accumulate=True)
        ref = base.double() + gy.double() @ w.double()
        scale = base.double().abs() + gy.double().abs() @ w.double().abs()
        assert ((acc.double() - ref).abs() / scale).max().item() < 4e-6
    y2 = torch.empty(T, N, device="cuda")
    ext.gemm_split(x, wf, None, y2, T, K, N, relu=True)
    assert (y2 >= 0).all() and _rel_err(torch.where(y2 > 0, y2, (y - bias).clamp(max=0)), x, w, None) < 4e-6


def test_narrow_and_short_problems_stay_on_the_library():
    """pointnet2_utils._gemm_nt / _gemm_nn: below 4096 tokens, for outputs narrower than 128 or K not a multiple of 32 the
    library GEMM runs (and gives the same numbers to f32 rounding)."""
    from pdanet_amd import pointnet2_utils as pu
    g = torch.Generator("cuda").manual_seed(3)
    for T, K, N in ((100, 64, 256), (10000, 48, 256), (10000, 64, 32), (10000, 64, 256)):
        x = torch.randn(T, K, device="cuda", generator=g); w = torch.randn(N, K, device="cuda", generator=g); b = torch.randn(N, device="cuda", generator=g)
        y = pu._gemm_nt(x, w, b, relu=True)
        assert torch.allclose(y, torch.relu(torch.nn.functional.linear(x, w, b)), atol=1e-4, rtol=1e-4)
        gx = pu._gemm_nn(y, w)
        assert torch.allclose(gx, y @ w, atol=2e-3, rtol=1e-4)


@pytest.mark.parametrize("groups,ns,K,N", [(2048, 16, 256, 512), (1031, 32, 512, 384), (777, 64, 256, 256), (5, 16, 64, 128)])
def test_maxpool_epilogue_equals_gemm_then_max(groups, ns, K, N):
    """pda_gemm_split_maxpool (the max over nsample in the epilogue of the last SA contraction, inference) against pda_gemm_split
    followed by the max: the same arithmetic, bit for bit, including a ragged last tile and an odd number of 128-column chunks."""
    from pdanet_amd import pointnet2_batch_cuda as ext
    T = groups * ns
    g = torch.Generator("cuda").manual_seed(groups + ns)
    x = torch.randn(T, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) * 0.1
    bias = torch.randn(N, device="cuda", generator=g)
    wf = ext.linear_split_pack(w, N, K)
    y = torch.empty(T, N, device="cuda")
    ext.gemm_split(x, wf, bias, y, T, K, N, relu=True)
    want = y.view(groups, ns, N).amax(dim=1)
    out = torch.full((groups, N), float("nan"), device="cuda")
    ext.gemm_split_maxpool(x, wf, bias, out, T, K, N, ns, relu=True)
    assert torch.equal(out, want)
    with pytest.raises(Exception):
        ext.gemm_split_maxpool(x, wf, bias, out, T, K, N, 24)


@pytest.mark.parametrize("B,N,M,ns,c1,c2", [(2, 2048, 1024, 32, 256, 256), (1, 700, 130, 16, 256, 512), (2, 1024, 200, 64, 512, 384)])
def test_gather_in_operand_load_equals_gather_then_gemm(B, N, M, ns, c1, c2):
    """pda_gemm_split_gather (inference: the first SA layer's output formed in the operand load of the second contraction)
    against pda_sa_point_gather followed by pda_gemm_split: the same arithmetic, bit for bit."""
    from pdanet_amd import pointnet2_batch_cuda as ext, pointnet2_utils as pu, synth
    g = torch.Generator("cuda").manual_seed(B + N + ns)
    xyz = torch.from_numpy(synth.batch_xyz(B, N, config_id=6)).cuda()
    ctr = (xyz[:, :M] + 0.1).contiguous()
    idx = pu.ball_query(8.4, ns, xyz, ctr)
    rows = torch.randn(B * N, c1, device="cuda", generator=g)
    w1 = torch.randn(c1, 3 + 64, device="cuda", generator=g) * 0.2
    b1 = torch.randn(c1, device="cuda", generator=g) * 0.3
    w2 = torch.randn(c2, c1, device="cuda", generator=g) * 0.1
    b2 = torch.randn(c2, device="cuda", generator=g)
    T = B * M * ns
    y1 = torch.empty(T, c1, device="cuda")
    ext.sa_point_gather(rows, xyz, ctr, idx, w1, y1, B, N, M, ns, c1, bias=b1, relu=True)
    wf = ext.linear_split_pack(w2, c2, c1)
    want = torch.empty(T, c2, device="cuda")
    ext.gemm_split(y1, wf, b2, want, T, c1, c2, relu=True)
    got = torch.full((T, c2), float("nan"), device="cuda")
    ext.gemm_split_gather(rows, xyz, ctr, idx, w1, b1, wf, b2, got, B, N, M, ns, c1, c2, relu=True)
    assert torch.equal(got, want)
