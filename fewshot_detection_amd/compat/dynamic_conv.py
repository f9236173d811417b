"""Drop-in alias: lets callers written against the reference (`from dynamic_conv import ...`, e.g.
train_meta.py:21-24, valid_ensemble.py:1-3) resolve to the MI355X implementation when this
directory is put on PYTHONPATH.  See INTEGRATION.md."""
from fewshot_detection_amd.dynamic_conv import *  # noqa: F401,F403
from fewshot_detection_amd import dynamic_conv as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
