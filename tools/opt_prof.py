import sys, time, torch, cProfile, pstats, os
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, ROOT)
import bench
from speaker_embedding_torch_amd.Modules import GE2E, GE2E_Loss
from speaker_embedding_torch_amd.Optim import FusedClipAdamW
hp = bench.Load_Hyper_Parameters(os.path.join(ROOT, 'speaker_embedding_torch_amd', 'Hyper_Parameters.yaml'))
dev = torch.device('cuda')
model = GE2E(hp, precision='bf16', seed=1234).to(dev); crit = GE2E_Loss().to(dev)
opt = FusedClipAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-6, max_norm=1.0)
model.train()
xs = [bench.synth_mel(960, 80, 160, 1234 + i, dev) for i in range(2)]
def step(i):
    emb = model(xs[i & 1]); loss = crit(emb, 15); opt.zero_grad(); loss.backward(); opt.step(); return loss
for i in range(5): step(i)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(20): step(i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumtime').print_stats(18)
