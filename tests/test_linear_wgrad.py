"""csrc/wgrad.hip (weight/bias gradient over a long token axis) against a float64 reference.
fp32 MFMA accumulation over up to 1M tokens: within 3e-5 of the largest entry."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tokens,n_in,n_out,bias", [(32768, 256, 128, True), (40000, 12, 32, False), (131072, 512, 1536, True),
                                                     (33000, 260, 256, False), (65536, 16, 8, True), (262144, 128, 256, True),
                                                     (50000, 64, 4, True), (300, 32, 64, True),
                                                     # wgrad_skinny_kernel: the layer-0 and position-MLP shapes, ragged token counts
                                                     (1048576, 4, 32, True), (1048576, 32, 64, True), (524288, 16, 16, False),
                                                     (32307, 12, 32, True), (64394, 64, 64, True), (8193, 36, 60, True),
                                                     # wgrad_split_kernel (both widths whole 256-column tiles): ragged token counts, 1-12 tiles
                                                     (62517, 512, 256, True), (70001, 256, 768, False), (131072, 256, 256, True),
                                                     (48365, 512, 1536, True), (100003, 512, 512, True),
                                                     # short token axes (64-row slices, one workgroup per CU) and a half-empty 128 x 128 tile
                                                     (4096, 256, 512, True), (8192, 64, 128, True), (4096, 12, 64, True),
                                                     # split form at its smallest token counts: a last slice of one row, a last step of one row
                                                     (4097, 512, 1536, True), (7953, 768, 512, False), (30529, 256, 256, True), (4099, 128, 64, False), (5000, 512, 1536, True)])
def test_gradients(tokens, n_in, n_out, bias):
    from pdanet_amd import pointnet2_utils as pu
    torch.manual_seed(tokens % 1000 + n_in)
    x = torch.randn(tokens, n_in, device="cuda", requires_grad=True)
    w = (torch.randn(n_out, n_in, device="cuda") * 0.1).requires_grad_(True)
    b = torch.randn(n_out, device="cuda", requires_grad=True) if bias else None
    go = torch.randn(tokens, n_out, device="cuda")
    y = pu.LinearLongTokens.apply(x, w, b)
    g = torch.autograd.grad(y, [x, w] + ([b] if bias else []), go)
    gw_ref = go.double().t() @ x.detach().double()
    # forward and input gradient: f32 GEMMs (the library, or csrc/gemm_split.hip from 4096 tokens): error at the level of
    # an f32 fmaf chain, relative to sum |x||w| -- a second f32 GEMM of another summation order is not the yardstick
    xd, wd = x.detach().double(), w.detach().double()
    y_ref = xd @ wd.t() + (b.detach().double() if bias else 0)
    assert ((y.double() - y_ref).abs() / (xd.abs() @ wd.abs().t() + (b.detach().double().abs() if bias else 0) + 1e-30)).max().item() < 2e-6
    assert (g[1].double() - gw_ref).abs().max().item() < 3e-5 * gw_ref.abs().max().item()
    assert ((g[0].double() - go.double() @ wd).abs() / (go.double().abs() @ wd.abs() + 1e-30)).max().item() < 2e-6
    if bias:
        gb_ref = go.double().sum(0)
        assert (g[2].double() - gb_ref).abs().max().item() < 3e-5 * gb_ref.abs().max().item() + 1e-4


def test_dispatch_threshold_and_views():
    from pdanet_amd import pointnet2_utils as pu
    conv = torch.nn.Conv2d(128, 256, 1, bias=False).cuda()
    x = torch.randn(2, 4096, 16, 128, device="cuda", requires_grad=True)       # 131072 tokens
    assert pu.LinearLongTokens.supported(torch.randn(131072, 32, device="cuda"), torch.randn(64, 32, device="cuda"))      # narrow: streaming form
    assert not pu.LinearLongTokens.supported(torch.randn(131072, 32, device="cuda"), torch.randn(96, 32, device="cuda"))  # in between: library
    assert not pu.LinearLongTokens.supported(torch.randn(2048, 32, device="cuda"), torch.randn(64, 32, device="cuda"))
    assert pu.LinearLongTokens.supported(x, conv.weight.flatten(1))
    assert not pu.LinearLongTokens.supported(x[:, :100], conv.weight.flatten(1))     # 3200 tokens: library path
    y = pu.linear(x, conv.weight.flatten(1), None)
    (gw,) = torch.autograd.grad(y.square().sum(), conv.weight)
    ref = torch.autograd.grad(F.linear(x, conv.weight.flatten(1)).square().sum(), conv.weight)[0]
    assert gw.shape == conv.weight.shape
    assert (gw - ref).abs().max().item() < 3e-5 * ref.abs().max().item()


@pytest.mark.parametrize("rows,cols", [(1, 8), (1000, 24), (65536, 768), (262144, 128), (4099, 1536), (33, 2048)])
def test_colsum_bf16(rows, cols):
    """pda_colsum_bf16 (dense-bf16 bias gradient) against a float64 column sum of the same bf16 values."""
    from pdanet_amd import pointnet2_batch_cuda as ext
    torch.manual_seed(rows + cols)
    g = (torch.randn(rows, cols, device="cuda") * 0.5 + 0.1).bfloat16()
    out = torch.empty(cols, device="cuda")
    ext.colsum_bf16(g, out, rows, cols)
    ref = g.double().sum(0)
    assert (out.double() - ref).abs().max().item() <= 2e-6 * g.double().abs().sum(0).max().item() + 1e-6
    out2 = torch.empty(cols, device="cuda")
    ext.colsum_bf16(g, out2, rows, cols)
    assert torch.equal(out, out2)            # fixed summation order
