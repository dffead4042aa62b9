"""Developer probe: time the flow train step (forward + backward + Adam + weight prep) at a bench config."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipoke_amd import configs
from ipoke_amd.flow import SupervisedMacowTransformer
from ipoke_amd.optim import FusedAdamAmsgrad

z = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
t0 = time.time()
m = SupervisedMacowTransformer(configs.flow_arch(z), dtype=dtype, max_batch=B, device="cuda")
print(f"built in {time.time()-t0:.1f}s, params {m.engine.n_params/1e9:.3f} B, ops {m.engine.n_ops}", flush=True)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(B, z, 8, 8, device="cuda", generator=g)
cond = torch.randn(B, 128, 8, 8, device="cuda", generator=g)
with torch.no_grad():
    m(x, cond)                                   # data-dependent init
    for name, p in m.named_parameters():
        if name.endswith("weight_g"):
            p.fill_(0.05)
m.mark_weights_updated()
opt = FusedAdamAmsgrad(m, lr=1e-4, weight_decay=1e-5)
m.train()
def step():
    out, logdet = m(x, cond)
    loss = (0.5 * (out ** 2).sum(dim=[1, 2, 3])).mean() - logdet.mean()
    loss.backward()
    opt.step()
    return loss
for i in range(3):
    l = step()
torch.cuda.synchronize()
print("warm loss", l.item(), flush=True)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
tt = {"fwd": 0, "bwd": 0, "opt": 0}
t0 = time.time()
for i in range(steps):
    evs[0].record()
    th = time.time()
    out, logdet = m(x, cond)
    host_fwd = time.time() - th
    loss = (0.5 * (out ** 2).sum(dim=[1, 2, 3])).mean() - logdet.mean()
    evs[1].record()
    th = time.time()
    loss.backward()
    host_bwd = time.time() - th
    evs[2].record()
    opt.step()
    evs[3].record()
    torch.cuda.synchronize()
    tt["fwd"] += evs[0].elapsed_time(evs[1]); tt["bwd"] += evs[1].elapsed_time(evs[2]); tt["opt"] += evs[2].elapsed_time(evs[3])
wall = (time.time() - t0) / steps * 1e3
print(f"z={z} B={B} {dtype}: wall {wall:.1f} ms/step; gpu fwd {tt['fwd']/steps:.1f} bwd {tt['bwd']/steps:.1f} opt+prep {tt['opt']/steps:.1f} ms; loss {loss.item():.3f}")
print(f"host enqueue fwd {host_fwd*1e3:.1f} ms bwd {host_bwd*1e3:.1f} ms")
print(f"mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
