"""Hyper-parameter bag with the reference's surface (utils/hparams.py:
``HParams(**kv).parse(['k','v',...]|dict|str).values()``, attribute access returns
None for unknown keys, ``update`` merges and returns self)."""
import ast


class HParams(object):
    def __init__(self, **init_hparams):
        object.__setattr__(self, 'keyvals', dict(init_hparams))

    def __getitem__(self, key):
        return self.keyvals.get(key)

    def __getattribute__(self, attribute):
        if attribute == '__dict__':
            return object.__getattribute__(self, 'keyvals')
        return object.__getattribute__(self, attribute)

    def __getattr__(self, key):
        return object.__getattribute__(self, 'keyvals').get(key)

    def __setattr__(self, key, value):
        self.keyvals[key] = value

    def update(self, values_dict):
        self.keyvals.update(values_dict)
        return self

    def parse(self, values):
        """dict -> merged; list/set ['k1','v1','k2','v2'] -> values literal-eval'ed
        when possible (utils/hparams.py:48-63); str -> literal dict."""
        if type(values) == dict:
            return self.update(values)
        if type(values) in (set, list, tuple):
            values = list(values)
            tmp = {}
            for k, v in zip(values[::2], values[1::2]):
                try:
                    tmp[k] = ast.literal_eval(v)
                except (ValueError, SyntaxError):
                    tmp[k] = v
            return self.update(tmp)
        return self.update(ast.literal_eval(values))

    def values(self):
        return self.keyvals

    def __str__(self):
        return str(self.keyvals)
