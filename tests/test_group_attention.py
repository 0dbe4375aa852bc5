"""GPU parity of csrc/group_attention.hip against torch's scaled_dot_product_attention (fp32).

The attention is floating point, so the bar is a tolerance: 2e-5 absolute on outputs of O(1)
inputs, gradients within 1e-4 of the largest gradient magnitude (fp32 MFMA accumulation order differs
from the framework's kernels)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(qkv, heads):
    G, S, D3 = qkv.shape
    hd = D3 // (3 * heads)
    q, k, v = qkv.view(G, S, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = q.double(), k.double(), v.double()
    p = torch.softmax(q @ k.transpose(-1, -2) / hd ** 0.5, dim=-1)
    return (p @ v).transpose(1, 2).reshape(G, S, heads * hd)


@pytest.mark.parametrize("S", [8, 16, 32])
@pytest.mark.parametrize("hd", [32, 64, 128])
@pytest.mark.parametrize("G,heads", [(1, 1), (5, 4), (1027, 4), (3, 3)])
def test_forward_backward(S, hd, G, heads):
    from pdanet_amd import pointnet2_utils as pu
    torch.manual_seed(S * 1000 + hd + G)
    qkv = (torch.randn(G, S, 3 * heads * hd, device="cuda") * 0.7).requires_grad_(True)
    assert pu.GroupAttention.supported(qkv, heads)
    out = pu.group_attention(qkv, heads)
    ref = _ref(qkv, heads)
    assert out.shape == ref.shape
    assert (out.double() - ref).abs().max().item() < 2e-5
    go = torch.randn_like(out)
    (g,) = torch.autograd.grad(out, qkv, go)
    (gr,) = torch.autograd.grad(ref, qkv, go.double())
    scale = gr.abs().max().item()
    assert (g.double() - gr.double()).abs().max().item() < 1e-4 * max(scale, 1.0)


def test_unsupported_shapes_fall_back():
    from pdanet_amd import pointnet2_utils as pu, pointnet2_batch_cuda as ext
    from pdanet_amd._lib import PdaError
    qkv = torch.randn(4, 12, 3 * 4 * 64, device="cuda")
    assert not pu.GroupAttention.supported(qkv, 4)
    out = torch.empty(4, 12, 256, device="cuda"); lse = torch.empty(4, 4, 12, device="cuda")
    with pytest.raises(PdaError):
        ext.group_attention_fwd(qkv, out, lse, 4, 12, 4, 64)


def test_transformer_layer_matches_sdpa_path():
    from pdanet_amd import pointnet2_modules as pm
    torch.manual_seed(3)
    layer = pm.TransformerEncoderLayerPreNorm(d_model=256, nhead=4, dim_feedforward=512, dropout=0.0).cuda()
    x = torch.randn(300, 16, 256, device="cuda", requires_grad=True)
    outs, grads = [], []
    for flag in (True, False):
        pm.GROUP_ATTENTION_KERNEL = flag
        y = pm._transformer_batch_first(layer, x)
        (g,) = torch.autograd.grad(y.square().mean(), x)
        outs.append(y.detach()); grads.append(g)
    pm.GROUP_ATTENTION_KERNEL = True
    assert torch.allclose(outs[0], outs[1], atol=2e-5, rtol=1e-5)
    assert torch.allclose(grads[0], grads[1], atol=1e-8 + 1e-4 * grads[1].abs().max().item())


@pytest.mark.parametrize("S,hd,G,heads", [(16, 64, 1027, 4), (32, 128, 77, 4), (8, 32, 5, 3), (32, 64, 1, 1)])
def test_bf16_boundary_variant(S, hd, G, heads):
    """pda_group_attention_{fwd,bwd}_bf16: same arithmetic, bf16 tensors at the HBM boundary.  On inputs that ARE
    bf16 values the results must equal the fp32 kernels' results rounded once to bf16 (1 ulp = 2^-8 relative slack for
    results that sit on a rounding boundary and differ in the last fp32 bits)."""
    from pdanet_amd import pointnet2_batch_cuda as ext
    torch.manual_seed(S + hd + G)
    D = heads * hd
    qkv_b = (torch.randn(G, S, 3 * D, device="cuda") * 0.7).bfloat16()
    go_b = torch.randn(G, S, D, device="cuda").bfloat16()
    qkv, go = qkv_b.float(), go_b.float()
    out, lse = torch.empty(G, S, D, device="cuda"), torch.empty(G, heads, S, device="cuda")
    ext.group_attention_fwd(qkv, out, lse, G, S, heads, hd)
    dqkv = torch.empty_like(qkv)
    ext.group_attention_bwd(qkv, go, lse, dqkv, G, S, heads, hd)
    out_b, lse_b = torch.empty(G, S, D, device="cuda", dtype=torch.bfloat16), torch.empty_like(lse)
    ext.group_attention_fwd(qkv_b, out_b, lse_b, G, S, heads, hd)
    dqkv_b = torch.empty_like(qkv_b)
    ext.group_attention_bwd(qkv_b, go_b, lse_b, dqkv_b, G, S, heads, hd)
    assert torch.equal(lse, lse_b)
    for got, want in ((out_b, out), (dqkv_b, dqkv)):
        assert got.dtype == torch.bfloat16
        exact = (got == want.bfloat16()).float().mean().item()
        assert exact > 0.999, exact
        assert (got.float() - want).abs().max().item() <= 2.0 ** -8 * want.abs().max().item()
    with pytest.raises(RuntimeError):
        ext.group_attention_fwd(qkv_b, out, lse, G, S, heads, hd)   # mixed dtypes are refused
