"""Reflection / CLI helpers with the reference's surface (utils/generic_utils.py):
name -> object lookup used by train.py / eval.py to resolve ``--model``,
``--input_parser``, ``--label_parser`` (case-insensitive, instantiates classes with
HParams-parsed kwargs when ``params`` is given)."""
import importlib
import inspect
import logging
import logging.config
import os
import re
import sys

from .hparams import HParams

_ALIASES = {            # the reference's top-level module names -> this package
    'core.models': 'asr_study_amd.core.models',
    'core.layers': 'asr_study_amd.core.layers',
    'core.metrics': 'asr_study_amd.core.metrics',
    'core.ctc_utils': 'asr_study_amd.core.ctc_utils',
    'preprocessing.audio': 'asr_study_amd.preprocessing.audio',
    'preprocessing.text': 'asr_study_amd.preprocessing.text',
    'datasets': 'asr_study_amd.datasets',
}


def safe_mkdirs(path):
    os.makedirs(path, exist_ok=True)
    return path


def _resolve(module):
    module = _ALIASES.get(module, module)
    if module not in sys.modules:
        importlib.import_module(module)
    return module


def inspect_module(module, to_dict=True, regex=False):
    """Members defined in ``module`` itself (not imported into it).  With ``regex``
    the name is a prefix pattern over every loaded sub-module (``datasets*``)."""
    if regex:
        import importlib
        import pkgutil
        root = _resolve(module.rstrip('*.'))
        pkg = importlib.import_module(root)
        for info in pkgutil.iter_modules(getattr(pkg, '__path__', [])):
            importlib.import_module(root + '.' + info.name)
        pattern = re.compile(re.escape(root) + r'(\.|$)')
        modules = {k: v for k, v in list(sys.modules.items()) if pattern.match(k) and v}
    else:
        module = _resolve(module)
        modules = {module: sys.modules[module]}
    members = []
    for key, value in modules.items():
        members.extend(inspect.getmembers(
            value, lambda m, key=key: getattr(m, '__module__', None) == key))
    return dict(members) if to_dict else members


def get_from_module(module, name, params=None, regex=False):
    """Class / function / instance called ``name`` (case-insensitive) in ``module``;
    classes are instantiated with ``HParams().parse(params)`` when params is given."""
    if name is None or str(name).lower() == 'none':
        return None
    members = {k.lower().strip(): v for k, v in inspect_module(module, regex=regex).items()}
    try:
        member = members[name.lower().strip()]
    except KeyError:
        raise KeyError("%s not found in %s.\n Valid values are: %s"
                       % (name, module, ', '.join(members.keys())))
    if member is not None and params is not None and inspect.isclass(member):
        return member(**HParams().parse(params).values())
    return member


def ld2dl(ld):
    """list of dicts -> dict of lists (all dicts share their keys)."""
    return {k: [d[k] for d in ld] for k in ld[0]}


def check_ext(fname, ext):
    ext = ext if ext[0] == '.' else '.' + ext
    return os.path.splitext(fname)[1] == ext


def parse_nondefault_args(args, default_args, argv=None):
    """Arguments the user actually passed (differ from the defaults or were named
    on the command line), as HParams (utils/generic_utils.py:105-115)."""
    named = [a.split('-')[-1] for a in (sys.argv if argv is None else argv)
             if a.startswith('-')]
    args_default = {k: v for k, v in vars(default_args).items() if k not in named}
    nondefault = {k: v for k, v in vars(args).items()
                  if k not in args_default or args_default[k] != v}
    return HParams().parse(nondefault)


def setup_logging(default_path='logging.yaml', default_level=logging.INFO, env_key='LOG_CFG'):
    path = os.getenv(env_key, None) or default_path
    if os.path.exists(path):
        import yaml
        with open(path, 'rt') as f:
            logging.config.dictConfig(yaml.safe_load(f.read()))
    else:
        logging.basicConfig(level=default_level)
