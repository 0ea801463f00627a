# -*- coding: utf-8 -*-
"""Label parsers with the reference's surface (preprocessing/text.py).

Python-3 restatement of CharParser: the reference's _sanitize uses py2-only
string.maketrans / str.translate(None, ...) (text.py:94) and dict.iteritems
(:138).  simple_char_parser: a-z -> 0..25, ' ' -> 26, blank id 27 (28 classes).
"""
import string

import numpy as np

try:                                    # optional, like the reference's dependency
    from unidecode import unidecode
except ImportError:                     # ASCII-only fallback: strip combining marks
    import unicodedata

    def unidecode(text):
        return ''.join(c for c in unicodedata.normalize('NFKD', text)
                       if not unicodedata.combining(c)).encode('ascii', 'ignore').decode()

PUNCTUATIONS = "'""-,.!?:;"
ACCENTS = u'ãõçâêôáíóúàüóé'


class BaseParser(object):
    """Interface class for all parsers (text.py:13-33)."""

    def __call__(self, _input):
        return self.map(_input)

    def map(self, _input):
        pass

    def imap(self, _input):
        pass

    def is_valid(self, _input):
        pass


class CharParser(BaseParser):
    """Maps text to a character vocabulary (text.py:36-143).

    mode: 'space'|'s', 'accents'|'a', 'punctuation'|'p', 'digits'|'d',
    'sensitive'|'S', joined with '|', or 'all'.
    """

    def __init__(self, mode='space'):
        self._permitted_modes = {'sensitive': 'S', 'space': 's', 'accents': 'a',
                                 'punctuation': 'p', 'digits': 'd'}
        if mode == 'all':
            self.mode = list(self._permitted_modes.values())
        else:
            self.mode = []
            for m in mode.split('|'):
                try:
                    self.mode.append(self._permitted_modes[m])
                except KeyError:
                    if m not in self._permitted_modes.values():
                        raise ValueError('Unknown mode %s' % m)
                    self.mode.append(m)
        self._vocab, self._inv_vocab = self._gen_vocab()

    def map(self, txt, sanitize=True):
        if sanitize:
            txt = self._sanitize(txt)
        return np.array([self._vocab[c] for c in txt], dtype='int32')

    def imap(self, labels):
        return ''.join([self._inv_vocab[int(l)] for l in labels])

    def _sanitize(self, text):
        text = ' '.join(text.split())                       # duplicated spaces
        if 'd' not in self.mode:
            text = ''.join([c for c in text if not c.isdigit()])
        if 'a' not in self.mode:
            text = unidecode(text)
        if 'p' not in self.mode:
            text = text.translate(str.maketrans("-'", '  '))
            text = text.translate(str.maketrans('', '', string.punctuation))
        if 's' not in self.mode:
            text = text.replace(' ', '')
        if 'S' not in self.mode:
            text = text.lower()
        return text

    def is_valid(self, text):
        try:
            self.map(text, sanitize=False)
            return True
        except KeyError:
            return False

    def _gen_vocab(self):
        vocab = {chr(v + ord('a')): v for v in range(ord('z') - ord('a') + 1)}
        if 'a' in self.mode:
            for a in ACCENTS:
                vocab[a] = len(vocab)
        if 'S' in self.mode:
            for char in list(vocab.keys()):
                vocab[char.upper()] = len(vocab)
        if 's' in self.mode:
            vocab[' '] = len(vocab)
        if 'p' in self.mode:
            for p in PUNCTUATIONS:
                vocab[p] = len(vocab)
        if 'd' in self.mode:
            for num in range(10):
                vocab[str(num)] = len(vocab)
        inv_vocab = {v: k for (k, v) in vocab.items()}
        inv_vocab[len(inv_vocab)] = '<b>'                   # blank label
        return vocab, inv_vocab


simple_char_parser = CharParser()
complex_char_parser = CharParser(mode='s|p|a|d')
