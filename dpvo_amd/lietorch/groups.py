"""lietorch: the SE3 forward surface of the reference's `dpvo.lietorch` (dpvo/lietorch/groups.py:53-322)
on the HIP kernels of dpvo_amd/csrc/geom.hip.

Only what the inference path touches is provided: SE3 {Identity, exp, log, inv, mul, act (act4), retr, scale,
matrix, translation, view/indexing, stack/cat}.  SO3 / RxSO3 / Sim3, the backward ops and the CPU backend
are training / gradcheck-only in the reference and are out of scope.
Data layout: [..., 7] = (tx, ty, tz, qx, qy, qz, qw), float32, on the GPU.
"""
import numpy as np
import torch

from .. import _lib as L


def _broadcast(x, y):
    """dpvo/lietorch/broadcasting.py:10-31 (without materialising anything when shapes already match)."""
    if y is None:
        return (x.reshape(-1, x.shape[-1]).contiguous(),), x.shape[:-1]
    assert x.dim() == y.dim()
    out_shape = tuple(max(n, m) for n, m in zip(x.shape[:-1], y.shape[:-1]))
    for n, m in zip(x.shape[:-1], y.shape[:-1]):
        assert n == m or n == 1 or m == 1
    x1 = x.expand(out_shape + (x.shape[-1],)).reshape(-1, x.shape[-1]).contiguous()
    y1 = y.expand(out_shape + (y.shape[-1],)).reshape(-1, y.shape[-1]).contiguous()
    return (x1, y1), out_shape


def _f32cuda(*ts):
    L.require_cuda(*ts)
    out = []
    for t in ts:
        if t.dtype != torch.float32:
            raise L.DPVOHipError("lietorch SE3 ops are float32 only on this backend")
        out.append(t)
    return out


def _op1(name, x, out_dim):
    (x1,), shape = _broadcast(x, None)
    _f32cuda(x1)
    y = torch.empty(x1.shape[0], out_dim, dtype=torch.float32, device=x1.device)
    L.check(getattr(L.lib(), name)(L.ptr(x1), L.ptr(y), L.i64(x1.shape[0]), L.stream()), name)
    return y.view(tuple(shape) + (out_dim,))


def _op2(name, x, y, out_dim):
    (x1, y1), shape = _broadcast(x, y)
    _f32cuda(x1, y1)
    z = torch.empty(x1.shape[0], out_dim, dtype=torch.float32, device=x1.device)
    L.check(getattr(L.lib(), name)(L.ptr(x1), L.ptr(y1), L.ptr(z), L.i64(x1.shape[0]), L.stream()), name)
    return z.view(tuple(shape) + (out_dim,))


class LieGroup:
    def __init__(self, data):
        self.data = data

    def __repr__(self):
        return "{}: size={}, device={}, dtype={}".format(self.group_name, self.shape, self.device, self.dtype)

    @property
    def shape(self):
        return self.data.shape[:-1]

    @property
    def device(self):
        return self.data.device

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def tangent_shape(self):
        return self.data.shape[:-1] + (self.manifold_dim,)

    @classmethod
    def Identity(cls, *batch_shape, **kwargs):
        if isinstance(batch_shape[0], (tuple, list, torch.Size)):
            batch_shape = tuple(batch_shape[0])
        numel = int(np.prod(batch_shape))
        data = cls.id_elem.reshape(1, -1)
        if 'device' in kwargs:
            data = data.to(kwargs['device'])
        if 'dtype' in kwargs:
            data = data.type(kwargs['dtype'])
        data = data.repeat(numel, 1)
        return cls(data).view(tuple(batch_shape))

    @classmethod
    def IdentityLike(cls, G):
        return cls.Identity(G.shape, device=G.data.device, dtype=G.data.dtype)

    def vec(self):
        return self.data

    def detach(self):
        return self.__class__(self.data.detach())

    def view(self, dims):
        return self.__class__(self.data.view(tuple(dims) + (self.embedded_dim,)))

    def __getitem__(self, index):
        return self.__class__(self.data[index])

    def __setitem__(self, index, item):
        self.data[index] = item.data

    def to(self, *args, **kwargs):
        return self.__class__(self.data.to(*args, **kwargs))

    def cpu(self):
        return self.__class__(self.data.cpu())

    def cuda(self):
        return self.__class__(self.data.cuda())

    def unbind(self, dim=0):
        return [self.__class__(x) for x in self.data.unbind(dim=dim)]


class SE3(LieGroup):
    group_name = 'SE3'
    group_id = 3
    manifold_dim = 6
    embedded_dim = 7
    id_elem = torch.as_tensor([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0])

    @classmethod
    def exp(cls, x):
        """exponential map (groups.py:130-133 -> lietorch_backends.expm)"""
        return cls(_op1("dpvo_se3_exp", x, 7))

    def log(self):
        return _op1("dpvo_se3_log", self.data, 6)

    def inv(self):
        return SE3(_op1("dpvo_se3_inv", self.data, 7))

    def mul(self, other):
        return SE3(_op2("dpvo_se3_mul", self.data, other.data, 7))

    def retr(self, a):
        """Exp(a) * X (groups.py:151-154)"""
        return SE3.exp(a).mul(self)

    def act(self, p):
        if p.shape[-1] == 4:
            return _op2("dpvo_se3_act4", self.data, p, 4)
        if p.shape[-1] == 3:
            p4 = torch.cat([p, torch.ones_like(p[..., :1])], dim=-1)
            return _op2("dpvo_se3_act4", self.data, p4, 4)[..., :3]
        raise ValueError(p.shape)

    def matrix(self):
        I = torch.eye(4, dtype=self.dtype, device=self.device)
        I = I.view([1] * (len(self.data.shape) - 1) + [4, 4])
        return SE3(self.data[..., None, :]).act(I).transpose(-1, -2)

    def translation(self):
        p = torch.as_tensor([0.0, 0.0, 0.0, 1.0], dtype=self.dtype, device=self.device)
        p = p.view([1] * (len(self.data.shape) - 1) + [4, ])
        return self.act(p)

    def scale(self, s):
        t, q = self.data.split([3, 4], -1)
        t = t * s.unsqueeze(-1)
        return SE3(torch.cat([t, q], dim=-1))

    def __mul__(self, other):
        if isinstance(other, LieGroup):
            return self.mul(other)
        if isinstance(other, torch.Tensor):
            return self.act(other)
        return NotImplemented


def cat(group_list, dim):
    data = torch.cat([X.data for X in group_list], dim=dim)
    return group_list[0].__class__(data)


def stack(group_list, dim):
    data = torch.stack([X.data for X in group_list], dim=dim)
    return group_list[0].__class__(data)
