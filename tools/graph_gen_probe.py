import os, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import test_model_gpu as tm
from percivaltts_amd import optimizertts_wgan, ops
ops.deterministic(True)
with contextlib.redirect_stdout(io.StringIO()):
    cfg, voc, mod, crit, a, gw, cw, X, Y, al = tm.build('default')
    opt = optimizertts_wgan.OptimizerTTSWGAN(cfg, mod, errtype='WLSWGAN', critic=crit)
    opt.prepare()
opt.generator_updates = 26
Xd, Yd = tm.f32(X), tm.f32(Y)
ag = torch.Generator().manual_seed(17)
alphas = [torch.rand(Xd.shape[0], generator=ag).cuda() for _ in range(5)]
moving = [t for k, t in mod.kerasmodel.weights() if 'moving' in k]
def state():
    return [t.detach().clone() for t in (opt.critic_opti.flat.flat, opt.gen_opti.flat.flat)] + [t.detach().clone() for t in moving]
def load(st):
    opt.wait_updates()
    opt.critic_opti.flat.flat.copy_(st[0]); opt.gen_opti.flat.flat.copy_(st[1])
    for dst, src in zip(moving, st[2:]): dst.copy_(src)
    opt.critic_opti.flat.epoch += 1; opt.gen_opti.flat.epoch += 1
NR = int(os.environ.get('NR', '5'))
trace = []
for b in range(NR):
    st = state()
    lc = opt.critic_step(Xd, Yd, alphas[b]); gc = opt.critic_opti.flat.grad.detach().clone()
    st2 = state()
    lg = opt.generator_step(Xd, Yd); gg = opt.gen_opti.flat.grad.detach().clone()
    trace.append((st, float(lc), gc, st2, float(lg), gg))
print('eager', [(round(t[1], 5), round(t[4], 5)) for t in trace])
opt.cfg.train_wgan_hipgraph = True; opt.cfg.train_wgan_graph_split = True
opt._graphed('critic', Xd, Yd, alphas[0]); opt._graphed('generator', Xd, Yd)
for b, (st, lc, gc, st2, lg, gg) in enumerate(trace):
    load(st)
    lc_g = float(opt._graphed('critic', Xd, Yd, alphas[b])); opt.wait_updates(); torch.cuda.synchronize()
    e1 = float((opt.critic_opti.flat.grad - gc).norm() / gc.norm())
    off = 0
    for p_ in opt.critic_opti.flat.params:
        n = p_.numel(); a_ = opt.critic_opti.flat.grad[off:off+n]; b_ = gc[off:off+n]
        er = float((a_ - b_).norm() / (b_.norm() + 1e-30))
        if er > 1e-4 or not torch.isfinite(a_).all(): print('   critic param', tuple(p_.shape), 'err', er, 'finite', bool(torch.isfinite(a_).all()), 'max', float(a_.abs().max()))
        off += n
    load(st2)
    lg_g = float(opt._graphed('generator', Xd, Yd)); opt.wait_updates(); torch.cuda.synchronize()
    e2 = float((opt.gen_opti.flat.grad - gg).norm() / gg.norm())
    off = 0
    for p_ in opt.gen_opti.flat.params:
        n = p_.numel(); a_ = opt.gen_opti.flat.grad[off:off+n]; b_ = gg[off:off+n]
        er = float((a_ - b_).norm() / (b_.norm() + 1e-30))
        if er > 1e-4 or not torch.isfinite(a_).all(): print('   gen param', tuple(p_.shape), 'err', er, 'finite', bool(torch.isfinite(a_).all()), 'max', float(a_.abs().max()))
        off += n
    print(b, lc_g, lc, e1, '|', lg_g, lg, e2)
ops.deterministic(False)
