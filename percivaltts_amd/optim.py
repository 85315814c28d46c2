"""Keras-2.2 Adam over one flat parameter buffer per network (reference: keras.optimizers.Adam as configured at
optimizertts_wgan.py:145,172 and optimizertts.py:387).

    t += 1 ; lr_t = lr*sqrt(1-b2^t)/(1-b1^t) ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr_t m/(sqrt(v)+eps)

One kernel launch updates the whole network; the step counter lives on the device so the launch can be replayed
from a hipGraph.  Under data parallelism the flat gradient is all-reduced (sum) first and `gscale = 1/world`.
"""
import numpy as np
import torch

from . import ops
from .layers import FlatParams


class KerasAdam(object):
    def __init__(self, model, device, lr, beta_1, beta_2, epsilon=1e-7):
        self.flat = FlatParams(model, device)
        self.m = torch.zeros_like(self.flat.flat)
        self.v = torch.zeros_like(self.flat.flat)
        self.step_count = torch.zeros((), dtype=torch.int32, device=device)
        self.lr, self.beta_1, self.beta_2, self.epsilon = float(lr), float(beta_1), float(beta_2), float(epsilon)

    def zero_grad(self):
        self.flat.grad.zero_()

    def step(self, gscale=1.0):
        ops.adam_keras_step_(self.flat.flat, self.flat.grad, self.m, self.v, self.step_count,
                             self.lr, self.beta_1, self.beta_2, self.epsilon, gscale)
        self.flat.epoch += 1          # the weights changed behind torch's version counters

    def clip_weights(self, lo, hi):
        ops.weight_clip_(self.flat.flat, lo, hi)
        self.flat.epoch += 1

    # Keras optimizer.weights order: iterations, then all m, then all v
    def get_weights(self):
        return [np.array(int(self.step_count.item()), dtype=np.int64), self.m.cpu().numpy(), self.v.cpu().numpy()]

    def set_weights(self, ws):
        it, m, v = ws
        if np.asarray(m).shape != tuple(self.m.shape):
            raise ValueError('optimizer state of {} values does not fit {} parameters'.format(np.asarray(m).size, self.m.numel()))
        self.step_count.fill_(int(it))
        self.m.copy_(torch.as_tensor(np.asarray(m), dtype=torch.float32))
        self.v.copy_(torch.as_tensor(np.asarray(v), dtype=torch.float32))

    def save(self, fname):
        it, m, v = self.get_weights()
        np.savez(fname, iterations=it, m=m, v=v)

    def load(self, fname):
        with np.load(fname) as z:
            self.set_weights([z['iterations'], z['m'], z['v']])
