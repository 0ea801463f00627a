from .audio import MFCC, FBank, LogFbank, Raw
from .text import CharParser, simple_char_parser, complex_char_parser
