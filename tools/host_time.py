import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench, os
from speaker_embedding_torch_amd.Modules import GE2E, GE2E_Loss
from speaker_embedding_torch_amd.Optim import FusedClipAdamW
from speaker_embedding_torch_amd.Arg_Parser import Recursive_Parse
import yaml
hp = bench.Load_Hyper_Parameters(os.path.join('/root/repo', 'speaker_embedding_torch_amd', 'Hyper_Parameters.yaml'))
dev = torch.device('cuda')
model = GE2E(hp, precision='bf16', seed=1234).to(dev); crit = GE2E_Loss().to(dev)
opt = FusedClipAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-6, max_norm=1.0)
model.train()
xs = [bench.synth_mel(960, 80, 160, 1234 + i, dev) for i in range(2)]
def step(i):
    emb = model(xs[i & 1]); loss = crit(emb, 15); opt.zero_grad(); loss.backward(); opt.step(); return loss
for i in range(5): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/20:.3f} ms/step, total {1e3*(t2-t0)/20:.3f} ms/step")
# breakdown of host time per phase
import collections
acc = collections.Counter()
for i in range(20):
    a = time.perf_counter(); emb = model(xs[i & 1]); b = time.perf_counter(); loss = crit(emb, 15); c = time.perf_counter()
    opt.zero_grad(); d = time.perf_counter(); loss.backward(); e = time.perf_counter(); opt.step(); f = time.perf_counter()
    acc['fwd'] += b - a; acc['loss'] += c - b; acc['zero'] += d - c; acc['bwd'] += e - d; acc['opt'] += f - e
torch.cuda.synchronize()
print({k: round(1e3 * v / 20, 3) for k, v in acc.items()})
