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
