"""Run-to-run and deferred-vs-immediate differences of the generator / critic gradients at configs[1] size."""
import sys, os, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from percivaltts_amd import vocoders, modeltts_common, networks_critic, optimizertts_wgan, backend_hip, ops

class A: batch = 64; frames = 400; ctx = 601
cfg = bench.make_cfg(A)
dev = backend_hip.device()
voc = vocoders.VocoderPML(16000, 0.005, 65, 20)
with contextlib.redirect_stdout(io.StringIO()):
    mod = modeltts_common.DCNNF0SpecNoiseFeatures(601, voc, cfg)
    crit = networks_critic.Critic(voc, 601, cfg)
    opt = optimizertts_wgan.OptimizerTTSWGAN(cfg, mod, errtype='WLSWGAN', critic=crit)
    opt.prepare()
X, Y = bench.synthetic(64, 400, 601, 86, 65, 321, dev)
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
cps = opt.critic_opti.flat.params
for p in cps: p.requires_grad_(False)
def gen(deferred, flush):
    ops._WG_FLUSH_AT = flush
    opt.gen_opti.zero_grad()
    if deferred:
        with ops.deferred_weight_grads():
            t, _ = opt.generator_loss(X, Y, training=True); t.backward()
    else:
        t, _ = opt.generator_loss(X, Y, training=True); t.backward()
    torch.cuda.synchronize()
    return opt.gen_opti.flat.grad.detach().clone()
a1 = gen(False, 0); a2 = gen(False, 0); b0 = gen(True, 0); b3 = gen(True, 3); b3b = gen(True, 3)
print('immediate vs immediate', rel(a2, a1))
print('deferred(end) vs immediate', rel(b0, a1))
print('deferred(3) vs immediate', rel(b3, a1))
print('deferred(3) vs deferred(3)', rel(b3b, b3))
# per-parameter worst
flat = opt.gen_opti.flat
off = 0
worst = []
for (name, p) in mod.kerasmodel.weights():
    if not p.requires_grad: continue
for p in flat.params:
    n = p.numel()
    d = rel(b3[off:off + n], a1[off:off + n]) if float(a1[off:off + n].norm()) > 0 else 0.0
    worst.append((d, tuple(p.shape))); off += n
print(sorted(worst, reverse=True)[:6])
