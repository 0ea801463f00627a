"""Adam / SGD with global clipnorm, Keras-1.2.2 constructor surface
(train.py:133-137: ``SGD(lr, momentum, clipnorm)`` / ``Adam(lr, clipnorm)``).

Both run as two launches over the model's single flat parameter buffer: the global
L2 norm of (g + 2*l2*w) in float64, then the fused clip + update (csrc/optim.hip).
``opt.lr`` is a plain attribute (train.py:122 sets it through K.set_value)."""
import torch

from .. import ops


class Optimizer(object):
    def __init__(self, lr, clipnorm=0.0):
        self.lr = float(lr)
        self.clipnorm = float(clipnorm or 0.0)
        self.iterations = 0

    def bind(self, model):
        self.state = [torch.zeros_like(model.params) for _ in range(self.n_state)]

    def get_state(self):
        return [s.detach().cpu().numpy() for s in self.state], self.iterations

    def set_state(self, arrays, iterations):
        for s, a in zip(self.state, arrays):
            s.copy_(torch.as_tensor(a))
        self.iterations = int(iterations)


class Adam(Optimizer):
    n_state = 2

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-8, decay=0.0,
                 clipnorm=0.0, **kwargs):
        super(Adam, self).__init__(lr, clipnorm)
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon
        self.decay = float(decay or 0.0)

    def step(self, model):
        # Keras 1.2.2: lr *= 1 / (1 + decay * iterations), iterations counted BEFORE this update
        lr = self.lr / (1.0 + self.decay * self.iterations) if self.decay else self.lr
        self.iterations += 1
        # norm -> device-side veto if a recurrent kernel flagged a timeout -> clipped update
        ops.grad_norm(model.params, model.grads, model._segs_dev, model._nseg, model._norm)
        ops.optim_guard(model._norm, model.params.device, model.veto_flags())
        ops.adam_step(model.params, model.grads, self.state[0], self.state[1], model._segs_dev,
                      model._nseg, model._norm, self.clipnorm, lr, self.iterations, self.beta_1,
                      self.beta_2, self.epsilon)


class SGD(Optimizer):
    n_state = 1

    def __init__(self, lr=0.01, momentum=0.0, decay=0.0, nesterov=False, clipnorm=0.0,
                 **kwargs):
        super(SGD, self).__init__(lr, clipnorm)
        self.momentum = float(momentum)
        self.decay = float(decay or 0.0)
        if nesterov:
            raise NotImplementedError('SGD(nesterov=True)')

    def step(self, model):
        lr = self.lr / (1.0 + self.decay * self.iterations) if self.decay else self.lr
        self.iterations += 1
        ops.grad_norm(model.params, model.grads, model._segs_dev, model._nseg, model._norm)
        ops.optim_guard(model._norm, model.params.device, model.veto_flags())
        ops.sgd_step(model.params, model.grads, self.state[0], model._segs_dev, model._nseg,
                     model._norm, self.clipnorm, lr, self.momentum)


def get(name):
    name = name.lower()
    if name == 'adam':
        return Adam()
    if name == 'sgd':
        return SGD()
    raise ValueError('unknown optimizer %r' % name)
