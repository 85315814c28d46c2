import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import io, contextlib
import torch
import test_model_gpu as tm
from percivaltts_amd import ops, _hip
with contextlib.redirect_stdout(io.StringIO()):
    cfg, voc, mod, crit, a, gw, cw, X, Y, al = tm.build('default')
    from percivaltts_amd import optimizertts_wgan
    opt = optimizertts_wgan.OptimizerTTSWGAN(cfg, mod, errtype='WLSWGAN', critic=crit)
    opt.prepare()
Xd, Yd = tm.f32(X), tm.f32(Y)
log = []
orig_call = _hip.call
def call2(name, *args, **kw):
    orig_call(name, *args, **kw)
ops_orig = {}
def wrap(fname):
    f = getattr(ops, fname)
    def g(*a, **k):
        out = f(*a, **k)
        outs = out if isinstance(out, (tuple, list)) else [out]
        ins = [t for t in a if torch.is_tensor(t)]
        log.append((fname, [t.detach().clone() for t in ins], [None if t is None else t.detach().clone() for t in outs]))
        return out
    setattr(ops, fname, g)
for fn in ('_conv2d_bwd_raw', '_conv2d_fwd_raw', 'colsums', '_affine_act_bwd_raw'):
    wrap(fn)
runs = []
for it in range(10):
    log.clear()
    opt.gen_opti.zero_grad()
    for p in opt.critic_opti.flat.params: p.requires_grad_(False)
    l, _ = opt.generator_loss(Xd, Yd, training=True)
    l.backward()
    for p in opt.critic_opti.flat.params: p.requires_grad_(True)
    torch.cuda.synchronize()
    runs.append(list(log))
ref = runs[0]
for it in range(1, 10):
    cur = runs[it]
    assert len(cur) == len(ref)
    for ci, ((n0, i0, o0), (n1, i1, o1)) in enumerate(zip(ref, cur)):
        din = [float((x - y).abs().max()) for x, y in zip(i0, i1)]
        dout = [None if x is None else float((x - y).abs().max()) for x, y in zip(o0, o1)]
        if any(d and d > 1e-7 for d in dout if d is not None) or any(d > 1e-7 for d in din):
            print('run', it, 'call', ci, n0, 'in shapes', [tuple(t.shape) for t in i0], 'in diffs', din, 'out diffs', dout)
            break
    else:
        print('run', it, 'all calls identical')
