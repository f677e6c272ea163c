"""CPU restatement of the Mamba-2 block's arithmetic as a plain sequential recurrence.  TEST INFRASTRUCTURE ONLY.

*** parity unpinned ***: the reference delegates to the third-party `mamba_ssm` (Rev fork, unpinned,
requirements.txt:32) via wenet/transformer/mamba_att_wrapper.py:24-35,49 and mamba2_bidirectional.py:72-144; the
package is not in the tree, so this restates the published algorithm (Dao & Gu 2024; mamba_ssm 2.x Mamba2.forward).
Cross-checked (not pinned: it is not the reference's own dependency) against an independent port of the same block that
IS installed -- transformers' Mamba2Mixer, same parameter names and layout, chunked-SSD CPU path -- with shared weights:
max |delta| 1.4e-6 on outputs of magnitude ~5 (tests/test_oracle_goldens.py)."""
import torch
import torch.nn.functional as F


def mamba2_forward(u, sd, p, headdim=64, d_state=128, d_conv=4):
    g = lambda n: sd[p + n].float()
    B, L, _ = u.shape
    d_inner = g("out_proj.weight").shape[1]
    H = d_inner // headdim
    zxbcdt = F.linear(u.float(), g("in_proj.weight"))
    z, xBC, dt = torch.split(zxbcdt, [d_inner, d_inner + 2 * d_state, H], dim=-1)
    xBC = F.silu(F.conv1d(xBC.transpose(1, 2), g("conv1d.weight"), g("conv1d.bias"), padding=d_conv - 1,
                          groups=xBC.shape[-1])[..., :L].transpose(1, 2))
    x, Bm, Cm = torch.split(xBC, [d_inner, d_state, d_state], dim=-1)
    dt = F.softplus(dt + g("dt_bias"))
    A = -torch.exp(g("A_log"))
    x = x.view(B, L, H, headdim)
    h = torch.zeros(B, H, headdim, d_state)
    ys = []
    for t in range(L):
        a = torch.exp(dt[:, t] * A)                                                    # (B, H)
        h = h * a[:, :, None, None] + (dt[:, t, :, None] * x[:, t])[..., None] * Bm[:, t, None, None, :]
        ys.append(torch.einsum("bhpn,bn->bhp", h, Cm[:, t]) + g("D")[None, :, None] * x[:, t])
    y = torch.stack(ys, 1).reshape(B, L, d_inner)
    y = y * F.silu(z)
    y = y * torch.rsqrt(y.pow(2).mean(-1, keepdim=True) + 1e-5) * g("norm.weight")
    return F.linear(y, g("out_proj.weight"))


def mamba2_bidirectional(u, sd, p, **kw):
    return (mamba2_forward(u, sd, p + "mamba_forward.", **kw)
            + torch.flip(mamba2_forward(torch.flip(u, [1]), sd, p + "mamba_backward.", **kw), [1])) / 2
