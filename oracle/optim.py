"""Oracle: Keras 1.2.2 Adam / SGD with global clipnorm.  TEST INFRASTRUCTURE ONLY.

Reference call site: train.py:133-137 -- ``Adam(lr, clipnorm=400)`` or
``SGD(lr, momentum, clipnorm)``.  The optimisers live in keras==1.2.2
(msc.yaml:89), not under /root/reference: PARITY UNPINNED by the reference.
Recalled semantics:

* clipnorm: norm = sqrt(sum over ALL gradient tensors of sum(g^2)); every g is
  scaled by clipnorm/norm when norm >= clipnorm (Keras ``clip_norm`` uses
  ``K.switch(n >= c, g*c/n, g)``);
* Adam (beta1 .9, beta2 .999, eps 1e-8, decay 0): t += 1;
  lr_t = lr * sqrt(1-beta2^t) / (1-beta1^t); m = b1 m + (1-b1) g;
  v = b2 v + (1-b2) g^2; p -= lr_t * m / (sqrt(v) + eps);
* SGD (momentum mu, no nesterov, decay 0): v = mu v - lr g; p += v.
"""
import numpy as np


def global_norm(grads):
    return float(np.sqrt(sum(np.sum(np.asarray(g, np.float64) ** 2)
                             for g in grads)))


def clip_by_global_norm(grads, clipnorm):
    if not clipnorm or clipnorm <= 0:
        return list(grads), global_norm(grads)
    n = global_norm(grads)
    if n >= clipnorm:
        return [g * (clipnorm / n) for g in grads], n
    return list(grads), n


class Adam(object):
    def __init__(self, lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-8,
                 clipnorm=0.0):
        self.lr, self.b1, self.b2, self.eps = lr, beta_1, beta_2, epsilon
        self.clipnorm = clipnorm
        self.t = 0
        self.m = self.v = None

    def step(self, params, grads):
        """params, grads: lists of arrays; params updated in place."""
        if self.m is None:
            self.m = [np.zeros_like(p) for p in params]
            self.v = [np.zeros_like(p) for p in params]
        grads, _ = clip_by_global_norm(grads, self.clipnorm)
        self.t += 1
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for p, g, m, v in zip(params, grads, self.m, self.v):
            g = g.astype(p.dtype)
            m[...] = self.b1 * m + (1.0 - self.b1) * g
            v[...] = self.b2 * v + (1.0 - self.b2) * g * g
            p[...] = p - lr_t * m / (np.sqrt(v) + self.eps)


class SGD(object):
    def __init__(self, lr=1e-2, momentum=0.9, clipnorm=0.0):
        self.lr, self.mu, self.clipnorm = lr, momentum, clipnorm
        self.vel = None

    def step(self, params, grads):
        if self.vel is None:
            self.vel = [np.zeros_like(p) for p in params]
        grads, _ = clip_by_global_norm(grads, self.clipnorm)
        for p, g, v in zip(params, grads, self.vel):
            v[...] = self.mu * v - self.lr * g.astype(p.dtype)
            p[...] = p + v
