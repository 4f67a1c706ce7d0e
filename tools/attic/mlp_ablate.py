import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from pdecontrolgym_amd.policy import FusedMLP
B = int(os.environ.get("MLP_B", 4096))
L = torch.nn.Linear
def seq(*m): return torch.nn.Sequential(*m).cuda()
T = torch.nn.Tanh
nets = {
 "257-64-64-1 tanh": seq(L(257,64),T(),L(64,64),T(),L(64,1),T()),
 "257-64-64-1 linear": seq(L(257,64),L(64,64),L(64,1)),
 "257-64 linear": seq(L(257,64)),
 "257-64 tanh": seq(L(257,64),T()),
 "64-64 linear": seq(L(64,64)),
 "64-1 linear": seq(L(64,1)),
 "8-64 linear": seq(L(8,64)),
 "1025-64 linear": seq(L(1025,64)),
 "257-256-256-1 relu (SAC actor shape)": seq(L(257,256),torch.nn.ReLU(),L(256,256),torch.nn.ReLU(),L(256,1),T()),
 "257-128-128-1 tanh": seq(L(257,128),T(),L(128,128),T(),L(128,1)),
}
for name, net in nets.items():
    fm = FusedMLP(net)
    x = torch.randn(B, fm.in_dim, device="cuda"); out = torch.zeros(B, fm.out_dim, device="cuda")
    fm.forward_into(x, out); torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fm.forward_into(x, out)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(200): fm.forward_into(x, out)
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    print(f"{name:40s} {(time.perf_counter()-t0)/5/200*1e6:7.2f} us")
