"""The identity behind sed_conv3x3_dgrad_poolstats (include/sed_hip.h), checked in float64 on the CPU against autograd.

ConvBlock's tail is y = avg_pool2d(relu(bn(z)), 2) (/root/reference/models/spectogram_models.py:156-158).  With
g = dL/d(bn output, pre-ReLU) = relu'(.) * up(dy)/4 the BatchNorm backward needs sum g and sum g*xhat per channel.  Both are
sums over POOLED pixels of dy times quantities the forward can keep at pooled resolution: the count of active pixels and the
pooled activation itself."""
import torch


def test_backward_statistics_from_pooled_tensors_equal_the_per_pixel_sums():
    torch.manual_seed(0)
    B, C, H, W = 3, 5, 9, 12                        # odd height: the pooling floor drops the last row
    z = torch.randn(B, C, H, W, dtype=torch.float64)
    gamma = torch.randn(C, dtype=torch.float64) + 1.5
    gamma[1] = -gamma[1]
    beta = torch.randn(C, dtype=torch.float64) * 0.5
    mean = z.mean(dim=(0, 2, 3))
    var = z.var(dim=(0, 2, 3), unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    xhat = (z - mean[None, :, None, None]) * invstd[None, :, None, None]
    bn = (gamma[None, :, None, None] * xhat + beta[None, :, None, None]).requires_grad_(True)
    y = torch.nn.functional.avg_pool2d(torch.relu(bn), 2)
    dy = torch.randn_like(y)
    (g,) = torch.autograd.grad(y, bn, dy)                         # the per-pixel gradient the reference's autograd forms
    s_ref = g.sum(dim=(0, 2, 3))
    q_ref = (g * xhat).sum(dim=(0, 2, 3))

    act = (bn.detach() > 0).to(torch.float64)
    cnt = torch.nn.functional.avg_pool2d(act, 2) * 4               # active pixels per pooled pixel
    s_new = (dy * cnt).sum(dim=(0, 2, 3)) / 4
    q_new = (dy * (y.detach() - beta[None, :, None, None] * cnt / 4)).sum(dim=(0, 2, 3)) / gamma
    torch.testing.assert_close(s_new, s_ref, rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(q_new, q_ref, rtol=1e-10, atol=1e-10)
