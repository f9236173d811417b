"""Drop-in alias: lets callers written against the reference (`from cfg import ...`, e.g.
train_meta.py:21-24, valid_ensemble.py:1-3) resolve to the MI355X implementation when this
directory is put on PYTHONPATH.  See INTEGRATION.md."""
from fewshot_detection_amd.cfg import *  # noqa: F401,F403
from fewshot_detection_amd import cfg as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

# the reference spells these with leading double underscores (cfg.py:7, 70, 152, 157)
__C = _impl.cfg
__configure_data, __configure_net, __configure_meta = _impl._configure_data, _impl._configure_net, _impl._configure_meta
