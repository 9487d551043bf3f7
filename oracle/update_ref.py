"""CPU restatement of the update operator -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows Update.forward (reference dpvo/net.py:74-92), GatedResidual / SoftAgg (dpvo/blocks.py:15-48) and the
torch_scatter 2.1.2 composite ops they call (`scatter_softmax` = max / sub / exp / sum / div, `scatter_sum`),
with the dtype flow autocast imposes at dpvo/dpvo.py:332: nn.Linear = f16 operands, wide
accumulate, one rounding to f16; LayerNorm -> f32; residual adds in f32; relu / sigmoid keep dtype.

torch_scatter is NOT vendored by the reference (environment.yml:12) and no reference test pins SoftAgg:
"parity unpinned" for that arithmetic; tests/golden/make_golden.py cross-checks this file against the
reference's own Update module imported with a scatter stub.

All math is done in float64 on values rounded to the dtype the reference would hold at that point.
"""
import numpy as np
import torch


def _h(x):
    """round to float16, keep float64 container"""
    return x.to(torch.float16).to(torch.float64)


def _f(x):
    return x.to(torch.float32).to(torch.float64)


def _linear(x, W, b):
    return _h(_h(x) @ _h(W).t() + _h(b))


def _layernorm(x, g, b, eps=1e-3):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return _f((x - mu) / torch.sqrt(var + eps) * g.double() + b.double())


def _groups(keys):
    """torch.unique(keys, return_inverse=True)[1]  (blocks.py:41)"""
    _, inv = np.unique(keys.numpy(), return_inverse=True)
    return torch.from_numpy(inv.astype(np.int64))


def _scatter_max(src, idx, n):
    out = torch.full((n, src.shape[1]), -float("inf"), dtype=src.dtype)
    return out.scatter_reduce(0, idx[:, None].expand_as(src), src, "amax", include_self=True)


def _scatter_sum(src, idx, n):
    out = torch.zeros(n, src.shape[1], dtype=src.dtype)
    return out.index_add(0, idx, src)


def soft_agg(x, keys, Wf, bf, Wg, bg, Wh, bh, half_scatter=True):
    """SoftAgg.forward (blocks.py:40-48).  half_scatter=True rounds every torch_scatter intermediate to f16
    (the reference feeds f16 tensors to torch_scatter, which computes in the input dtype)."""
    r = _h if half_scatter else (lambda t: t)
    jx = _groups(keys)
    n = int(jx.max()) + 1
    gx = _linear(x, Wg, bg)
    fx = _linear(x, Wf, bf)
    mx = _scatter_max(gx, jx, n)
    ex = r(torch.exp(r(gx - mx[jx])))
    sm = r(_scatter_sum(ex, jx, n))
    w = r(ex / sm[jx])
    y = r(_scatter_sum(r(fx * w), jx, n))
    return _linear(y, Wh, bh)[jx]


def gated_residual(x, sd, prefix):
    gate = _h(torch.sigmoid(_linear(x, sd[prefix + "gate.0.weight"], sd[prefix + "gate.0.bias"])))
    res = _linear(torch.relu(_linear(x, sd[prefix + "res.0.weight"], sd[prefix + "res.0.bias"])),
                  sd[prefix + "res.2.weight"], sd[prefix + "res.2.bias"])
    return _f(x + _h(gate * res))


def neighbors_np(kk, jj):
    from . import neighbors
    return neighbors(kk, jj)


def update_forward(sd, net, inp, corr, ii, jj, kk, half_scatter=True):
    """sd: state dict of the Update module (keys without the 'update.' prefix), float32 tensors on CPU.
    net [E,384] (f32 or f16 values), inp [E,384] f16 values, corr [E,882] f16 values; ii,jj,kk int64 [E].
    Returns net [E,384] f64 (f32-representable), delta [E,2], weight [E,2] (f16-representable)."""
    sd = {k: v.detach().cpu() for k, v in sd.items()}
    net = net.detach().cpu().double(); inp = _h(inp.detach().cpu()); corr = _h(corr.detach().cpu())
    ii, jj, kk = ii.cpu().long(), jj.cpu().long(), kk.cpu().long()

    c = _linear(corr, sd["corr.0.weight"], sd["corr.0.bias"])
    c = torch.relu(c)
    c = _linear(c, sd["corr.2.weight"], sd["corr.2.bias"])
    c = _layernorm(c, sd["corr.3.weight"], sd["corr.3.bias"])
    c = torch.relu(c)
    c = _linear(c, sd["corr.5.weight"], sd["corr.5.bias"])
    net = _f(_f(net + inp) + c)                                           # net.py:77 (f32 adds)
    net = _layernorm(net, sd["norm.weight"], sd["norm.bias"])             # :78

    ix, jx = neighbors_np(kk.numpy(), jj.numpy())                         # :80
    ix = torch.from_numpy(ix); jx = torch.from_numpy(jx)
    mask_ix = (ix >= 0).double()[:, None]; mask_jx = (jx >= 0).double()[:, None]
    t = mask_ix * net[ix]                                                 # negative index wraps, then masked (:81-84)
    t = _linear(torch.relu(_linear(t, sd["c1.0.weight"], sd["c1.0.bias"])), sd["c1.2.weight"], sd["c1.2.bias"])
    net = _f(net + t)
    t = mask_jx * net[jx]
    t = _linear(torch.relu(_linear(t, sd["c2.0.weight"], sd["c2.0.bias"])), sd["c2.2.weight"], sd["c2.2.bias"])
    net = _f(net + t)

    for name, keys in (("agg_kk", kk), ("agg_ij", ii * 12345 + jj)):      # :87-88
        a = soft_agg(net, keys, sd[name + ".f.weight"], sd[name + ".f.bias"], sd[name + ".g.weight"],
                     sd[name + ".g.bias"], sd[name + ".h.weight"], sd[name + ".h.bias"], half_scatter)
        net = _f(net + a)

    net = _layernorm(net, sd["gru.0.weight"], sd["gru.0.bias"])           # :90
    net = gated_residual(net, sd, "gru.1.")
    net = _layernorm(net, sd["gru.2.weight"], sd["gru.2.bias"])
    net = gated_residual(net, sd, "gru.3.")

    r = torch.relu(net)
    delta = _linear(r, sd["d.1.weight"], sd["d.1.bias"])                  # :92
    weight = _h(torch.sigmoid(_linear(r, sd["w.1.weight"], sd["w.1.bias"])))
    return net, delta, weight
