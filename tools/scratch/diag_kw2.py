import sys, os, importlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
aof = ge.load_package()
synth = importlib.import_module("aero_optical_flow_amd.synth")
from oracle import pyoracle as orc
kw = dict(pyramid_levels=2, mean_subtract=1, tile=16, search=8, value_threshold=12000)
W, H = 192, 160
p = aof.default_params(W, H, **kw)
po = orc.params_from(p)
dev = torch.device("cuda:0")
def check(tag, blocks, flows, hp, hc):
    gb, gf = aof.blocks_view(blocks), aof.flows_view(flows)
    bad = []
    for i in range(hp.shape[0]):
        ref = orc.flow_pair(po, hp[i], hc[i])
        if gb[i].tobytes() != ref["blocks"].tobytes() or gf[i].tobytes() != ref["flow"].tobytes():
            bad.append(i)
    print(tag, "bad pairs:", bad)
for n in (4, 10):
    for seed in (5200, 5300, 5301):
        hp, hc, _ = synth.make_batch(W, H, n, 9, seed, noise=3, brightness=5)
        eng = aof.FlowEngine(p, 0)
        b, f, _ = eng.flow_batch(torch.from_numpy(hp).to(dev), torch.from_numpy(hc).to(dev))
        torch.cuda.synchronize()
        check(f"eager n={n} seed={seed}", b, f, hp, hc)
        eng.set_split_coarse(True)
# graph: capture with seed 5200, replay with 5300
n = 10
hp, hc, _ = synth.make_batch(W, H, n, 9, 5200, noise=3, brightness=5)
prev, cur = torch.from_numpy(hp).to(dev), torch.from_numpy(hc).to(dev)
eng = aof.FlowEngine(p, 0)
L = aof.workspace_layout(p, n)
ws = torch.zeros(L.total_bytes, dtype=torch.uint8, device=dev)
b, f, _ = eng.flow_batch(prev, cur, workspace=ws)
torch.cuda.synchronize()
check("eager before capture", b, f, hp, hc)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    eng.flow_batch(prev, cur, blocks=b, flows=f, workspace=ws)
g.replay(); torch.cuda.synchronize()
check("replay same data", b, f, hp, hc)
hp2, hc2, _ = synth.make_batch(W, H, n, 9, 5300, noise=3, brightness=5)
prev.copy_(torch.from_numpy(hp2)); cur.copy_(torch.from_numpy(hc2))
g.replay(); torch.cuda.synchronize()
check("replay new data", b, f, hp2, hc2)
w = ws.cpu().numpy()
sums = w[L.sums:L.sums + 16 * n].view(np.uint32).reshape(n, 2, 2)
print("sums[0]", sums[0], "expected", hp2[0].sum(), hc2[0].sum())
f1 = aof.flows_view(torch.from_numpy(w[L.l1_flows:L.l1_flows + 16 * n].reshape(n, 16)))
print("l1 flows", f1[:3])
eng.flow_batch(prev, cur, blocks=b, flows=f, workspace=ws); torch.cuda.synchronize()
check("eager new data", b, f, hp2, hc2)
w = ws.cpu().numpy()
sums = w[L.sums:L.sums + 16 * n].view(np.uint32).reshape(n, 2, 2)
print("sums[0]", sums[0])
