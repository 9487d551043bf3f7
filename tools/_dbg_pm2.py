import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpvo_amd import synthetic as S, net as N
from dpvo_amd.graph import GraphPlan
dev = torch.device("cuda:0")
torch.manual_seed(7)
upd = N.Update(3).to(dev)
ii, jj, kk = (t.to(dev) for t in S.replay_graph(40))
E = ii.numel()
g = torch.Generator().manual_seed(5)
net = torch.randn(E, 384, generator=g).to(dev)
imap = torch.randn(3456, 384, generator=g).half().to(dev)
corr = torch.zeros(E, 896, dtype=torch.float16, device=dev); corr[:, :882] = torch.randn(E, 882, generator=g).half().to(dev)
plan = GraphPlan(ii, jj, kk)
def run(fz):
    x, (d, w, _) = upd(net[None].clone(), imap[None], corr[None], None, ii, jj, kk, plan=plan, inp_rows=kk, inp_mod=3456, corr_is_padded=True, fused=fz)
    torch.cuda.synchronize()
    return x[0].clone(), d[0].clone(), w[0].clone()
ref = run(True)
outs = [run("pm2") for _ in range(4)]
print("status", upd.pm_status.view(torch.int32)[0].item())
for k, o in enumerate(outs):
    dx = (o[0] - ref[0]).abs().max(1).values
    print(k, "vs fused: max", dx.max().item(), "rows > 1e-2:", (dx > 1e-2).sum().item(), "nan rows", torch.isnan(o[0]).any(1).sum().item())
    if k:
        dd = (o[0] - outs[0][0]).abs().max(1).values
        bad = (dd > 0).nonzero().flatten()
        print("   vs run 0: differing rows", bad.numel(), "max", dd.max().item(), bad[:20].tolist())
        if bad.numel():
            print("   kk of differing rows", kk[bad[:20]].tolist())
