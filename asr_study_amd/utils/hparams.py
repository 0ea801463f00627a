"""Free-form hyper-parameter bag used by the command lines.

Behavioural contract (what train.py / eval.py of the reference rely on,
utils/hparams.py): construct from keyword arguments; ``parse`` accepts a dict, a flat
``[key, value, key, value, ...]`` list whose values are run through
``ast.literal_eval`` when they look like literals (otherwise kept as strings), or a
string holding a dict literal; ``update`` merges and returns the bag; attribute and
item access return ``None`` for unknown keys; ``values()`` exposes the dict;
``vars(bag)`` works (``__dict__`` is the dict)."""
import ast


def _literal(text):
    """'512' -> 512, '[1, 2]' -> [1, 2], 'tanh' -> 'tanh'."""
    if not isinstance(text, str):
        return text
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        return text


def _pairs(flat):
    flat = list(flat)
    return {flat[i]: _literal(flat[i + 1]) for i in range(0, len(flat) - 1, 2)}


class HParams(object):
    __slots__ = ('keyvals',)

    def __init__(self, **initial):
        object.__setattr__(self, 'keyvals', dict(initial))

    # -- merging -----------------------------------------------------------
    def update(self, mapping):
        self.keyvals.update(mapping)
        return self

    def parse(self, spec):
        if isinstance(spec, dict):
            return self.update(spec)
        if isinstance(spec, (list, tuple, set)):
            return self.update(_pairs(spec))
        return self.update(ast.literal_eval(spec))

    # -- access ------------------------------------------------------------
    def values(self):
        return self.keyvals

    def __getitem__(self, key):
        return self.keyvals.get(key)

    def __getattribute__(self, name):
        if name == '__dict__':
            return object.__getattribute__(self, 'keyvals')
        return object.__getattribute__(self, name)

    def __getattr__(self, name):           # only reached for unknown attributes
        return object.__getattribute__(self, 'keyvals').get(name)

    def __setattr__(self, name, value):
        self.keyvals[name] = value

    def __contains__(self, key):
        return key in self.keyvals

    def __repr__(self):
        return 'HParams(%r)' % (self.keyvals,)

    __str__ = lambda self: str(self.keyvals)
