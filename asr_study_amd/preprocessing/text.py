# -*- coding: utf-8 -*-
"""Character-level label parsers (the reference's ``preprocessing/text.py`` surface:
``CharParser(mode)``, ``.map`` / ``.imap`` / ``.is_valid`` / ``__call__``, and the two
ready-made instances ``simple_char_parser`` / ``complex_char_parser``).

Vocabulary order (it fixes the class ids the CTC layer is trained on): a-z = 0..25,
then -- when enabled -- the Brazilian-Portuguese accented letters, upper-case copies,
space, punctuation, digits; the CTC blank is the id after the last symbol.  With the
default mode ('space') that is 27 symbols + blank = 28 classes.

Sanitising pipeline of ``map`` (same order as the reference's ``_sanitize``, which is
observable: e.g. digits are removed *after* runs of spaces were collapsed): collapse
whitespace, drop digits, transliterate accents (unidecode), turn '-' and "'" into
spaces and drop the remaining punctuation, drop spaces, lower-case -- each step only
when the corresponding mode letter is absent."""
import string
import unicodedata

import numpy as np

try:
    from unidecode import unidecode as _to_ascii
except ImportError:                      # optional dependency, as in the reference
    def _to_ascii(text):
        decomposed = unicodedata.normalize('NFKD', text)
        return ''.join(ch for ch in decomposed if not unicodedata.combining(ch)) \
            .encode('ascii', 'ignore').decode()

PUNCTUATIONS = "'""-,.!?:;"
ACCENTS = u'ãõçâêôáíóúàüóé'
MODE_LETTERS = {'sensitive': 'S', 'space': 's', 'accents': 'a', 'punctuation': 'p',
                'digits': 'd'}
_DASHES_TO_SPACE = str.maketrans("-'", '  ')
_NO_PUNCT = str.maketrans('', '', string.punctuation)


class BaseParser(object):
    def __call__(self, _input):
        return self.map(_input)

    def map(self, _input):
        raise NotImplementedError

    def imap(self, _input):
        raise NotImplementedError

    def is_valid(self, _input):
        raise NotImplementedError


def _parse_mode(mode):
    if mode == 'all':
        return sorted(MODE_LETTERS.values())
    letters = []
    for token in mode.split('|'):
        if token in MODE_LETTERS:
            letters.append(MODE_LETTERS[token])
        elif token in MODE_LETTERS.values():
            letters.append(token)
        else:
            raise ValueError('Unknown mode %s' % token)
    return letters


def _build_vocab(flags):
    symbols = list(string.ascii_lowercase)
    if 'a' in flags:
        symbols += list(ACCENTS)
    if 'S' in flags:
        symbols += [c.upper() for c in symbols]
    if 's' in flags:
        symbols.append(' ')
    if 'p' in flags:
        symbols += list(PUNCTUATIONS)
    if 'd' in flags:
        symbols += list(string.digits)
    vocab = {}
    for sym in symbols:                  # later duplicates keep the LATER id (dict assign)
        vocab[sym] = len(vocab)
    inverse = {idx: sym for sym, idx in vocab.items()}
    inverse[len(inverse)] = '<b>'        # CTC blank
    return vocab, inverse


class CharParser(BaseParser):
    def __init__(self, mode='space'):
        self._permitted_modes = dict(MODE_LETTERS)
        self.mode = _parse_mode(mode)
        self._vocab, self._inv_vocab = _build_vocab(self.mode)

    def _sanitize(self, text):
        flags = self.mode
        text = ' '.join(text.split())
        if 'd' not in flags:
            text = ''.join(ch for ch in text if not ch.isdigit())
        if 'a' not in flags:
            text = _to_ascii(text)
        if 'p' not in flags:
            text = text.translate(_DASHES_TO_SPACE).translate(_NO_PUNCT)
        if 's' not in flags:
            text = text.replace(' ', '')
        return text if 'S' in flags else text.lower()

    def map(self, txt, sanitize=True):
        chars = self._sanitize(txt) if sanitize else txt
        return np.fromiter((self._vocab[c] for c in chars), dtype='int32', count=len(chars))

    def imap(self, labels):
        return ''.join(self._inv_vocab[int(i)] for i in labels)

    def is_valid(self, text):
        return all(c in self._vocab for c in text)


simple_char_parser = CharParser()
complex_char_parser = CharParser(mode='s|p|a|d')
