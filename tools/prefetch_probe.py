"""Where the time of the PCIe-inclusive leg goes: waiting for the prefetcher vs the device step, and the loader's own pace."""
import sys, os, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from percivaltts_amd import vocoders, modeltts_common, networks_critic, optimizertts_wgan, backend_hip, data

class A: batch = 64; frames = 400; ctx = 601
cfg = bench.make_cfg(A)
cfg.train_wgan_parallel_streams = True
dev = backend_hip.device()
voc = vocoders.VocoderPML(16000, 0.005, 65, 20)
with contextlib.redirect_stdout(io.StringIO()):
    mod = modeltts_common.DCNNF0SpecNoiseFeatures(601, voc, cfg)
    crit = networks_critic.Critic(voc, 601, cfg)
    opt = optimizertts_wgan.OptimizerTTSWGAN(cfg, mod, errtype='WLSWGAN', critic=crit)
    opt.prepare()
opt.generator_updates = 26
X, Y = bench.synthetic(64, 400, 601, 86, 65, 123, dev)
pool = [(X.cpu().numpy(), Y.cpu().numpy())] * 3
for i in range(6): opt.device_step(i, X, Y)
torch.cuda.synchronize()
# loader alone
pf = data.BatchPrefetcher(lambda i: pool[i % 3], 20, device=dev, depth=2)
t0 = time.time(); n = 0
for x, y in pf: n += 1
torch.cuda.synchronize(); print('loader alone: ms per batch', (time.time() - t0) / n * 1e3)
# resident
t0 = time.time()
for i in range(18): opt.device_step(i, X, Y)
torch.cuda.synchronize(); print('resident: ms per step', (time.time() - t0) / 18 * 1e3)
# with prefetcher
pf = data.BatchPrefetcher(lambda i: pool[i % 3], 18, device=dev, depth=2)
tw = ts = 0.0; t0 = time.time(); it = iter(pf); i = 0
while True:
    a = time.time()
    try: x, y = next(it)
    except StopIteration: break
    b = time.time(); opt.device_step(i, x, y); c = time.time(); tw += b - a; ts += c - b; i += 1
torch.cuda.synchronize(); print('prefetched: ms per step', (time.time() - t0) / 18 * 1e3, 'waiting', tw / 18 * 1e3, 'enqueue', ts / 18 * 1e3)
# plain synchronous .to(device) per step
t0 = time.time()
for i in range(18):
    x = torch.from_numpy(pool[0][0]).to(dev); y = torch.from_numpy(pool[0][1]).to(dev)
    opt.device_step(i, x, y)
torch.cuda.synchronize(); print('synchronous .to(): ms per step', (time.time() - t0) / 18 * 1e3)
