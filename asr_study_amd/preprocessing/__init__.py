"""Host-side mirrors of the reference's feature extractors and label parsers; the
feature math itself runs in csrc/frontend.hip."""
from . import audio, text
from .audio import Feature, FBank, MFCC, LogFbank, Raw  # noqa: F401
from .text import (CharParser, complex_char_parser,  # noqa: F401
                   simple_char_parser)

__all__ = ['audio', 'text', 'Feature', 'FBank', 'MFCC', 'LogFbank', 'Raw', 'CharParser',
           'simple_char_parser', 'complex_char_parser']
