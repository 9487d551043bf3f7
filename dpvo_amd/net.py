"""VONet / Update / Patchifier with the reference's module tree and state-dict keys (dpvo/net.py:27-184,
dpvo/blocks.py:15-48), so `dpvo.pth` loads unchanged, and `Update.forward` executed by the seven HIP kernels of
dpvo_amd/csrc/update_fused.hip instead of ~60 torch / torch_scatter launches.  (`fused=False` / `composite=False`
run the launch-by-launch comparator implementation of libdpvo_hip_cmp.so instead: tests and measurements only.)

The nn.Module parameters are the single source of truth; `Update.pack()` derives the f16 operand images the
kernels consume (what autocast's per-call weight casts produce in the reference, dpvo/dpvo.py:332).
"""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import workspace
from . import altcorr
from .extractor import BasicEncoder4
from .graph import GraphPlan
from .utils import coords_grid_with_index

DIM = 384

# bench.py sets this to a list to collect (start_event, end_event, n_edges) of every Update.forward: HIP events recorded on the
# launch stream around the operator (MFMA roofline measurement); None = no overhead.
PROFILE = None

EPI_NONE, EPI_RELU, EPI_SIGMOID, EPI_RESADD, EPI_GATED, EPI_RELU_SIG = range(6)
# Update.forward runs the seven row-tile kernels of update_fused.hip -- always the same path, whatever the box (round 2's
# run-time autotune between five candidates is gone: a tracker's output must not depend on a timing).  The comparator
# (launch-by-launch update.hip) is reached explicitly: Update.forward(..., fused=False).
FUSED_DEFAULT = True


# ------------------------------------------------------------------------------------------ kernels' Python face
def linear(A, W, bias, out=None, epilogue=EPI_NONE, rows=None, gate=None, n_split=0, K=None, out16=None):
    """dpvo_linear: A [M,>=K] f16/f32 (row stride = A.stride(0)), W [N,K] f16, bias [N] f16."""
    M = A.shape[0]
    N = W.shape[0]
    K = K or W.shape[1]
    assert A.stride(1) == 1 and W.stride(1) == 1
    if epilogue in (EPI_RESADD, EPI_GATED):
        assert out is not None and out.dtype == torch.float32
    elif out is None:
        out = torch.empty(M, N, dtype=torch.float16, device=A.device)
    L.check(L.cmp_lib().dpvo_linear(L.ptr(A), L.i32(L.dtype_code(A.dtype)), L.i64(A.stride(0)), L.ptr(rows), L.ptr(W),
                                L.i64(W.stride(0)), L.ptr(bias), L.ptr(out), L.i64(out.stride(0)), L.ptr(gate),
                                L.i64(gate.stride(0) if gate is not None else 0), L.ptr(out16),
                                L.i64(out16.stride(0) if out16 is not None else 0), L.i32(epilogue), L.i32(n_split),
                                L.i64(M), L.i32(N), L.i32(K), L.stream()), "dpvo_linear")
    return out


def layernorm(x, gamma, beta, eps=1e-3, add1=None, add1_rows=None, add1_mod=0, add2=None, y_f32=None, y_f16=None,
              relu_f16=False):
    M, D = x.shape
    assert x.is_contiguous()
    L.check(L.cmp_lib().dpvo_layernorm(L.ptr(x), L.i32(L.dtype_code(x.dtype)), L.ptr(add1), L.ptr(add1_rows),
                                   L.i64(add1_mod), L.ptr(add2), L.ptr(gamma), L.ptr(beta), L.f32(eps), L.ptr(y_f32),
                                   L.ptr(y_f16), L.i32(1 if relu_f16 else 0), L.i64(M), L.i32(D), L.stream()),
            "dpvo_layernorm")


def softagg(fg, perm, off, n_groups_dev, n_groups):
    y = torch.empty(max(n_groups, 1), DIM, dtype=torch.float16, device=fg.device)
    L.check(L.lib().dpvo_softagg(L.ptr(fg), L.i64(fg.stride(0)), L.ptr(perm), L.ptr(off), L.ptr(n_groups_dev),
                                 L.i64(n_groups), L.ptr(y), L.i32(DIM), L.stream()), "dpvo_softagg")
    return y


def gather_add(net, hy, group, net16=None):
    L.check(L.cmp_lib().dpvo_gather_add(L.ptr(net), L.ptr(hy), L.ptr(group), L.ptr(net16), L.i64(net.shape[0]), L.i32(DIM),
                                    L.stream()), "dpvo_gather_add")


def heads(net, Wd, bd, Ww, bw, coords=None, target_out=None, weight_out=None):
    """delta, weight (net.py:92); with `coords` [.,E,2,P,P] also target = coords[..., P//2, P//2] + delta (dpvo.py:340)
    written to `target_out` [E,2]; `weight_out` [E,2] receives the weights in place"""
    E = net.shape[0]
    delta = torch.empty(E, 2, dtype=torch.float32, device=net.device)
    weight = weight_out if weight_out is not None else torch.empty(E, 2, dtype=torch.float32, device=net.device)
    if coords is not None:
        assert coords.is_contiguous() and coords.dtype == torch.float32 and target_out is not None
        L.check(L.cmp_lib().dpvo_heads_target(L.ptr(net), L.ptr(Wd), L.ptr(bd), L.ptr(Ww), L.ptr(bw), L.ptr(coords),
                                          L.i32(coords.shape[-1]), L.ptr(delta), L.ptr(weight), L.ptr(target_out), L.i64(E),
                                          L.i32(DIM), L.stream()), "dpvo_heads_target")
    else:
        L.check(L.cmp_lib().dpvo_heads(L.ptr(net), L.ptr(Wd), L.ptr(bd), L.ptr(Ww), L.ptr(bw), L.ptr(delta), L.ptr(weight),
                                   L.i64(E), L.i32(DIM), L.stream()), "dpvo_heads")
    return delta, weight


class _UpdParams(ctypes.Structure):
    """dpvo_update_params_t"""
    _fields_ = [(k, ctypes.c_void_p) for k in (
        "c0_w", "c0_b", "c2_w", "c2_b", "cln_g", "cln_b", "c5_w", "c5_b", "norm_g", "norm_b",
        "c1_w0", "c1_b0", "c1_w2", "c1_b2", "c2n_w0", "c2n_b0", "c2n_w2", "c2n_b2",
        "akk_wfg", "akk_bfg", "akk_wh", "akk_bh", "aij_wfg", "aij_bfg", "aij_wh", "aij_bh",
        "g0_g", "g0_b", "g0_wrg", "g0_brg", "g0_w2", "g0_b2", "g1_g", "g1_b", "g1_wrg", "g1_brg", "g1_w2", "g1_b2",
        "d_w", "d_b", "w_w", "w_b")]


UF_NLIN = 19          # DPVO_UF_* of include/dpvo_hip.h


class _UpdFusedParams(ctypes.Structure):
    """dpvo_update_fused_params_t"""
    _fields_ = [("w", ctypes.c_void_p * UF_NLIN), ("b", ctypes.c_void_p * UF_NLIN), ("ln_g", ctypes.c_void_p * 4),
                ("ln_b", ctypes.c_void_p * 4), ("d_w", ctypes.c_void_p), ("d_b", ctypes.c_void_p), ("w_w", ctypes.c_void_p),
                ("w_b", ctypes.c_void_p), ("tiling", ctypes.c_int32), ("start_skew", ctypes.c_int32)]


def fused_pack(W, K=None, chained=True):
    """dpvo_update_fused_pack: torch Linear weight [384, k] f16 -> the MFMA fragment image of update_fused.hip"""
    assert W.dtype == torch.float16 and W.shape[0] == DIM and W.stride(1) == 1
    K = K or W.shape[1]
    out = torch.empty(DIM * K, dtype=torch.float16, device=W.device)
    L.check(L.lib().dpvo_update_fused_pack(L.ptr(W), L.i64(W.stride(0)), L.i32(K), L.i32(W.shape[1]), L.i32(1 if chained else 0),
                                           L.ptr(out), L.stream()), "dpvo_update_fused_pack")
    return out


# ------------------------------------------------------------------------------------------ modules (reference tree)
class GradientClip(nn.Module):
    """identity in forward (dpvo/blocks.py:74-89); kept so that Sequential indices match the checkpoint keys"""

    def forward(self, x):
        return x


class GatedResidual(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.gate = nn.Sequential(nn.Linear(dim, dim), nn.Sigmoid())
        self.res = nn.Sequential(nn.Linear(dim, dim), nn.ReLU(inplace=True), nn.Linear(dim, dim))


class SoftAgg(nn.Module):
    def __init__(self, dim=512, expand=True):
        super().__init__()
        self.dim = dim
        self.expand = expand
        self.f = nn.Linear(self.dim, self.dim)
        self.g = nn.Linear(self.dim, self.dim)
        self.h = nn.Linear(self.dim, self.dim)


class Update(nn.Module):
    def __init__(self, p):
        super().__init__()
        self.c1 = nn.Sequential(nn.Linear(DIM, DIM), nn.ReLU(inplace=True), nn.Linear(DIM, DIM))
        self.c2 = nn.Sequential(nn.Linear(DIM, DIM), nn.ReLU(inplace=True), nn.Linear(DIM, DIM))
        self.norm = nn.LayerNorm(DIM, eps=1e-3)
        self.agg_kk = SoftAgg(DIM)
        self.agg_ij = SoftAgg(DIM)
        self.gru = nn.Sequential(
            nn.LayerNorm(DIM, eps=1e-3), GatedResidual(DIM),
            nn.LayerNorm(DIM, eps=1e-3), GatedResidual(DIM))
        self.corr = nn.Sequential(
            nn.Linear(2 * 49 * p * p, DIM), nn.ReLU(inplace=True), nn.Linear(DIM, DIM),
            nn.LayerNorm(DIM, eps=1e-3), nn.ReLU(inplace=True), nn.Linear(DIM, DIM))
        self.d = nn.Sequential(nn.ReLU(inplace=False), nn.Linear(DIM, 2), GradientClip())
        self.w = nn.Sequential(nn.ReLU(inplace=False), nn.Linear(DIM, 2), GradientClip(), nn.Sigmoid())
        self.ncorr = 2 * 49 * p * p
        self.kpad = (self.ncorr + 31) // 32 * 32
        self._packed = None
        # tile shape / soft start of the seven-launch kernels: per instance, handed to the library with every call (the
        # library keeps no state); -1 = the library's default tiling.  Bit-identical results for every value.
        self.tiling = -1
        self.start_skew = 0

    # -------------------------------------------------------------------------------------- operand images
    def pack(self):
        """f16 weight images for the kernels (re-run after loading / changing parameters)."""
        h = lambda t: t.detach().to(torch.float16).contiguous()
        f = lambda t: t.detach().float().contiguous()
        dev = self.norm.weight.device
        P = {}
        w0 = torch.zeros(DIM, self.kpad, dtype=torch.float16, device=dev)
        w0[:, :self.ncorr] = h(self.corr[0].weight)
        P["c0"] = (w0, h(self.corr[0].bias))
        P["c2"] = (h(self.corr[2].weight), h(self.corr[2].bias))
        P["cln"] = (f(self.corr[3].weight), f(self.corr[3].bias))
        P["c5"] = (h(self.corr[5].weight), h(self.corr[5].bias))
        P["norm"] = (f(self.norm.weight), f(self.norm.bias))
        for name, mod in (("c1", self.c1), ("c2n", self.c2)):
            P[name] = (h(mod[0].weight), h(mod[0].bias), h(mod[2].weight), h(mod[2].bias))
        for name, mod in (("akk", self.agg_kk), ("aij", self.agg_ij)):
            P[name] = (h(torch.cat([mod.f.weight, mod.g.weight], 0)), h(torch.cat([mod.f.bias, mod.g.bias], 0)),
                       h(mod.h.weight), h(mod.h.bias))
        for name, ln, gr in (("g0", self.gru[0], self.gru[1]), ("g1", self.gru[2], self.gru[3])):
            P[name] = (f(ln.weight), f(ln.bias),
                       h(torch.cat([gr.res[0].weight, gr.gate[0].weight], 0)),
                       h(torch.cat([gr.res[0].bias, gr.gate[0].bias], 0)),
                       h(gr.res[2].weight), h(gr.res[2].bias))
        P["d"] = (h(self.d[1].weight), h(self.d[1].bias))
        P["w"] = (h(self.w[1].weight), h(self.w[1].bias))
        # pointer table of dpvo_update_forward (field order = dpvo_update_params_t in include/dpvo_hip.h)
        tab = [P["c0"][0], P["c0"][1], P["c2"][0], P["c2"][1], P["cln"][0], P["cln"][1], P["c5"][0], P["c5"][1],
               P["norm"][0], P["norm"][1], *P["c1"], *P["c2n"], *P["akk"], *P["aij"], *P["g0"], *P["g1"], *P["d"], *P["w"]]
        assert len(tab) == len(_UpdParams._fields_)
        P["_params"] = _UpdParams(*[ctypes.c_void_p(t.data_ptr()) for t in tab])
        if dev.type == "cuda":
            # fragment images of the row-tile-resident kernels (update_fused.hip), order = DPVO_UF_* (include/dpvo_hip.h)
            gr0, gr1 = self.gru[1], self.gru[3]
            lins = [(self.corr[0], False), (self.corr[2], True), (self.corr[5], True),
                    (self.c1[0], True), (self.c1[2], True), (self.c2[0], True), (self.c2[2], True),
                    (self.agg_kk.f, True), (self.agg_kk.g, True), (self.agg_kk.h, True),
                    (self.agg_ij.f, True), (self.agg_ij.g, True), (self.agg_ij.h, True),
                    (gr0.gate[0], True), (gr0.res[0], True), (gr0.res[2], True),
                    (gr1.gate[0], True), (gr1.res[0], True), (gr1.res[2], True)]
            assert len(lins) == UF_NLIN
            P["_fw"] = [fused_pack(h(m.weight), K=(self.kpad if not ch else DIM), chained=ch) for m, ch in lins]
            P["_fb"] = [h(m.bias) for m, _ in lins]
            fp = _UpdFusedParams()
            for i in range(UF_NLIN):
                fp.w[i] = P["_fw"][i].data_ptr()
                fp.b[i] = P["_fb"][i].data_ptr()
            for i, name in enumerate(("cln", "norm", "g0", "g1")):
                fp.ln_g[i] = P[name][0].data_ptr()
                fp.ln_b[i] = P[name][1].data_ptr()
            fp.d_w, fp.d_b, fp.w_w, fp.w_b = (t.data_ptr() for t in (*P["d"], *P["w"]))
            fp.tiling, fp.start_skew = self.tiling, self.start_skew
            P["_fparams"] = fp
        self._packed = P
        return P

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = None
        return super().load_state_dict(*a, **k)

    # -------------------------------------------------------------------------------------- forward
    @torch.no_grad()
    def forward(self, net, inp, corr, flow, ii, jj, kk, plan=None, inp_rows=None, inp_mod=0, corr_is_padded=False,
                out=None, coords=None, target_out=None, weight_out=None, composite=True, fused=None,
                net_rows=None):
        """update operator (net.py:74-92).  net [1,E,384] f32/f16, inp [1,E,384] f16 (or, with `inp_rows`, the
        un-gathered imap [1,S,384] plus int64 row ids taken modulo inp_mod), corr [1,E,882] f16.
        `out` (optional f32 [E,384] buffer, may alias `net`): receives the new hidden state (in-place update).
        `coords` [1,E,2,P,P] + `target_out` / `weight_out` [E,2] f32 (optional): the heads kernel also writes
        target = coords[..., P//2, P//2] + delta (dpvo.py:340) and the weights straight into the caller's edge arrays.
        `net_rows` = (int64 device tensor, n_kept, buffer): the state of edge g is buffer[net_rows[g]] (buffer: the [capacity, 384]
        tensor `net` is the head of) for g < n_kept and zero after (a
        removal whose compaction of `net` was deferred, EdgeStore.keep(defer_net=True)); the seven-launch kernels gather it in
        their first kernel, every other path gathers it here first.
        Returns net f32 [1,E,384], (delta f32 [1,E,2], weight f32 [1,E,2], None)."""
        P = self._packed or self.pack()
        L.require_cuda(net, inp, corr, ii, jj, kk)
        E = ii.numel()
        dev = net.device
        if E == 0:
            z = torch.zeros(1, 0, 2, device=dev)
            return net.float(), (z, z.clone(), None)
        if plan is None:
            plan = GraphPlan(ii, jj, kk)

        corr2 = corr.reshape(E, -1)
        if corr_is_padded:
            # rows of a [E, kpad] f16 buffer whose columns >= 882 are zero (altcorr.corr_pyramid output)
            assert corr2.dtype == torch.float16 and corr2.stride(1) == 1 and corr2.stride(0) == self.kpad
        else:
            buf = torch.zeros(E, self.kpad, dtype=torch.float16, device=dev)
            buf[:, :corr2.shape[1]] = corr2
            corr2 = buf

        net2 = net.reshape(E, DIM)
        if net2.dtype != torch.float32 and composite:
            # the reference's hidden state is f16 until its first update (dpvo.py:220); here it is f32 from the start (same
            # values): a half tensor is promoted instead of being routed to another implementation
            net2 = net2.float()
        if not net2.is_contiguous():
            net2 = net2.contiguous()
        inp2 = inp.reshape(-1, DIM)
        if inp2.dtype != torch.float16:
            inp2 = inp2.half()
        inp2 = inp2.contiguous()

        if net_rows is not None:
            seven = composite and net2.dtype == torch.float32 and (fused is True or (fused is None and FUSED_DEFAULT))
            if not seven:
                rows, n_kept, src = net_rows
                tmp = workspace.get(E * DIM * 4, dev, "net_gather")[:E * DIM * 4].view(torch.float32).view(E, DIM)
                if n_kept:
                    torch.index_select(src.reshape(-1, DIM), 0, rows[:n_kept], out=tmp[:n_kept])
                tmp[n_kept:].zero_()
                net2, net_rows = tmp, None
        prof = PROFILE
        if prof is not None:
            ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
            ev0.record()
            res = self.forward_impl(net2, inp2, corr2, ii, jj, kk, plan, inp_rows, inp_mod, out, coords, target_out, weight_out,
                                    composite, fused, E, dev, P, net_rows)
            ev1.record()
            prof.append((ev0, ev1, E))
            return res
        return self.forward_impl(net2, inp2, corr2, ii, jj, kk, plan, inp_rows, inp_mod, out, coords, target_out, weight_out,
                                 composite, fused, E, dev, P, net_rows)

    def forward_impl(self, net2, inp2, corr2, ii, jj, kk, plan, inp_rows, inp_mod, out, coords, target_out, weight_out, composite,
                     fused, E, dev, P, net_rows=None):
        if composite and net2.dtype == torch.float32:
            # the whole operator as ONE library call (dpvo_update_forward issues the same launches as the code below;
            # `composite=False` keeps the launch-by-launch path for tests)
            x = out.reshape(E, DIM) if out is not None else torch.empty(E, DIM, dtype=torch.float32, device=dev)
            assert x.dtype == torch.float32 and x.is_contiguous()
            delta = torch.empty(E, 2, dtype=torch.float32, device=dev)
            weight = weight_out if weight_out is not None else torch.empty(E, 2, dtype=torch.float32, device=dev)
            if coords is not None:
                assert coords.is_contiguous() and coords.dtype == torch.float32 and target_out is not None
            maxg = max(plan.n_patches_host, plan.n_pairs_host)
            if fused is None:
                fused = FUSED_DEFAULT
            if fused:
                # seven launches of row-tile-resident kernels (update_fused.hip)
                P["_fparams"].tiling, P["_fparams"].start_skew = self.tiling, self.start_skew
                nbytes = L.lib().dpvo_update_fused_workspace_bytes(L.i64(E), L.i64(maxg))
                ws = workspace.get(nbytes, dev, "update_fused")
                rows, n_kept = net_rows[:2] if net_rows is not None else (None, 0)
                L.check(L.lib().dpvo_update_forward_fused_rows(
                    ctypes.byref(P["_fparams"]), L.ptr(net2), L.ptr(rows), L.i64(n_kept), L.ptr(inp2), L.ptr(inp_rows), L.i64(inp_mod),
                    L.ptr(corr2), L.i64(corr2.stride(0)), L.ptr(plan.buf), L.i64(plan.n_patches_host), L.i64(plan.n_pairs_host),
                    L.ptr(coords), L.i32(coords.shape[-1] if coords is not None else 0), L.ptr(x), L.ptr(delta),
                    L.ptr(weight), L.ptr(target_out if coords is not None else None), L.i64(E), L.ptr(ws),
                    ctypes.c_size_t(ws.numel()), L.stream()), "dpvo_update_forward_fused_rows")
                return x.view(1, E, DIM), (delta.view(1, E, 2), weight.view(1, E, 2), None)
            nbytes = L.cmp_lib().dpvo_update_workspace_bytes(L.i64(E), L.i64(maxg))
            ws = workspace.get(nbytes, dev, "update")
            L.check(L.cmp_lib().dpvo_update_forward(
                ctypes.byref(P["_params"]), L.ptr(net2), L.ptr(inp2), L.ptr(inp_rows), L.i64(inp_mod), L.ptr(corr2),
                L.i64(corr2.stride(0)), L.ptr(plan.buf), L.i64(plan.n_patches_host), L.i64(plan.n_pairs_host),
                L.ptr(coords), L.i32(coords.shape[-1] if coords is not None else 0), L.ptr(x), L.ptr(delta), L.ptr(weight),
                L.ptr(target_out if coords is not None else None), L.i64(E), L.ptr(ws), ctypes.c_size_t(ws.numel()),
                L.stream()), "dpvo_update_forward")
            return x.view(1, E, DIM), (delta.view(1, E, 2), weight.view(1, E, 2), None)

        # net = net + inp + self.corr(corr); net = self.norm(net)                              (net.py:77-78)
        h1 = linear(corr2, P["c0"][0], P["c0"][1], epilogue=EPI_RELU, K=self.kpad)
        h2 = linear(h1, P["c2"][0], P["c2"][1])
        layernorm(h2, P["cln"][0], P["cln"][1], y_f16=h1, relu_f16=True)
        c = linear(h1, P["c5"][0], P["c5"][1], out=h2)
        x = out.reshape(E, DIM) if out is not None else torch.empty(E, DIM, dtype=torch.float32, device=dev)
        assert x.dtype == torch.float32 and x.is_contiguous()
        x16 = torch.empty(E, DIM, dtype=torch.float16, device=dev)      # f16 operand image of `net`, kept in step
        layernorm(net2, P["norm"][0], P["norm"][1], add1=inp2, add1_rows=inp_rows, add1_mod=inp_mod, add2=c, y_f32=x,
                  y_f16=x16)

        # net = net + c1(mask_ix * net[:,ix]); net = net + c2(mask_jx * net[:,jx])              (net.py:80-85)
        for name, rows in (("c1", plan.ix), ("c2n", plan.jx)):
            W0, b0, W2, b2 = P[name]
            t = linear(x16, W0, b0, out=h1, epilogue=EPI_RELU, rows=rows)
            linear(t, W2, b2, out=x, epilogue=EPI_RESADD, out16=x16)

        # net = net + agg_kk(net, kk); net = net + agg_ij(net, ii*12345 + jj)                    (net.py:87-88)
        fg = torch.empty(E, 2 * DIM, dtype=torch.float16, device=dev)
        for name, perm, off, cnt_dev, cnt, grp, want16 in (
                ("akk", plan.perm_k, plan.patch_off, plan.counts[0:1], plan.n_patches_host, plan.ku, True),
                ("aij", plan.perm_p, plan.pair_off, plan.counts[1:2], plan.n_pairs_host, plan.pu, False)):
            Wfg, bfg, Wh, bh = P[name]
            linear(x16, Wfg, bfg, out=fg)
            y = softagg(fg, perm, off, cnt_dev, cnt)
            hy = linear(y, Wh, bh)
            gather_add(x, hy, grp, net16=x16 if want16 else None)

        # net = self.gru(net): 2 x (LayerNorm, x + gate(x) * res(x))                            (net.py:90)
        for name in ("g0", "g1"):
            g, b, Wrg, brg, W2, b2 = P[name]
            layernorm(x, g, b, y_f32=x, y_f16=h1)
            linear(h1, Wrg, brg, out=fg, epilogue=EPI_RELU_SIG, n_split=DIM)
            linear(fg[:, :DIM], W2, b2, out=x, epilogue=EPI_GATED, gate=fg[:, DIM:])

        delta, weight = heads(x, P["d"][0], P["d"][1], P["w"][0], P["w"][1], coords=coords,    # net.py:92
                              target_out=target_out, weight_out=weight_out)
        return x.view(1, E, DIM), (delta.view(1, E, 2), weight.view(1, E, 2), None)


class Patchifier(nn.Module):
    """Patch extraction (net.py:95-157).  In the tracker the encoders run through csrc/encoder.hip (dpvo_amd/encoders.py) and the
    gathers through dpvo_frame_patches; this module's forward is the API-parity path (torch convolutions) used by the tests."""

    def __init__(self, patch_size=3):
        super().__init__()
        self.patch_size = patch_size
        self.fnet = BasicEncoder4(output_dim=128, norm_fn='instance')
        self.inet = BasicEncoder4(output_dim=DIM, norm_fn='none')
        self._grid_cache = {}

    def __image_gradient(self, images):
        gray = ((images + 0.5) * (255.0 / 2)).sum(dim=2)
        dx = gray[..., :-1, 1:] - gray[..., :-1, :-1]
        dy = gray[..., 1:, :-1] - gray[..., :-1, :-1]
        g = torch.sqrt(dx ** 2 + dy ** 2)
        g = F.avg_pool2d(g, 4, 4)
        return g

    def forward(self, images, patches_per_image=80, disps=None, centroid_sel_strat='RANDOM', return_color=False,
                coords=None, half=False, images_f16=None, return_coords=False, maps=None):
        """`coords` ([n, patches, 2] float, optional) injects the patch centroids (deterministic tests / oracle
        replay); otherwise they are drawn exactly like the reference (x then y, net.py:131-133).
        `half`: the encoders hold f16 weights (DPVO casts them once) and are fed an f16 copy of the image."""
        if maps is not None:
            # (fmap, imap) already computed by the HIP encoders as NHWC f16 maps, "/ 4" included: expose NCHW views
            fmap = maps[0].permute(2, 0, 1)[None, None]
            imap = maps[1].permute(2, 0, 1)[None, None]
        else:
            enc_in = images_f16 if images_f16 is not None else (images.half() if half else images)
            fmap = self.fnet(enc_in) / 4.0
            imap = self.inet(enc_in) / 4.0
        b, n, c, h, w = fmap.shape
        P = self.patch_size
        dev = fmap.device

        if coords is None:
            if centroid_sel_strat == 'GRADIENT_BIAS':
                # in f32 whatever the encoders eat: gray reaches ~765 and dx ** 2 overflows f16 (the reference's image is
                # f32 here, net.py:104-111 -- autocast does not downcast sum / pow / sqrt)
                g = self.__image_gradient(images.float())
                x = torch.randint(1, w - 1, size=[n, 3 * patches_per_image], device=dev)
                y = torch.randint(1, h - 1, size=[n, 3 * patches_per_image], device=dev)
                coords = torch.stack([x, y], dim=-1).float()
                g = altcorr.patchify(g[0, :, None], coords, 0).view(n, 3 * patches_per_image)
                ix = torch.argsort(g, dim=1)
                x = torch.gather(x, 1, ix[:, -patches_per_image:])
                y = torch.gather(y, 1, ix[:, -patches_per_image:])
            elif centroid_sel_strat == 'RANDOM':
                x = torch.randint(1, w - 1, size=[n, patches_per_image], device=dev)
                y = torch.randint(1, h - 1, size=[n, patches_per_image], device=dev)
            else:
                raise NotImplementedError(f"Patch centroid selection not implemented: {centroid_sel_strat}")
            coords = torch.stack([x, y], dim=-1).float()

        imap = altcorr.patchify(imap[0], coords, 0).view(b, -1, DIM, 1, 1)
        gmap = altcorr.patchify(fmap[0], coords, P // 2).view(b, -1, 128, P, P)
        if return_color:
            clr = altcorr.patchify(images[0], 4 * (coords + 0.5), 0).view(b, -1, 3)
        if disps is None:
            key = (b, n, h, w, str(dev))
            grid = self._grid_cache.get(key)
            if grid is None:                      # (x, y, 1) grid: constant for a given frame size
                grid, _ = coords_grid_with_index(torch.ones(b, n, h, w, device=dev), device=dev)
                self._grid_cache = {key: grid}
        else:
            grid, _ = coords_grid_with_index(disps, device=dev)
        patches = altcorr.patchify(grid[0], coords, P // 2).view(b, -1, 3, P, P)
        index = None
        if not return_coords:
            index = torch.arange(n, device=dev).view(n, 1)
            index = index.repeat(1, patches_per_image).reshape(-1)
        if return_coords:
            return fmap, gmap, imap, patches, index, coords
        if return_color:
            return fmap, gmap, imap, patches, index, clr
        return fmap, gmap, imap, patches, index


class VONet(nn.Module):
    def __init__(self, use_viewer=False):
        super().__init__()
        self.P = 3
        self.patchify = Patchifier(self.P)
        self.update = Update(self.P)
        self.DIM = DIM
        self.RES = 4

    def forward(self, *a, **k):
        raise NotImplementedError("VONet.forward is the training loop of the reference (net.py:187-272): out of scope")
