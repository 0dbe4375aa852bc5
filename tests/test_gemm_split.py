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
                                   (33000, 512, 512), (5000, 384, 1536), (2048, 512, 2048)])
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
