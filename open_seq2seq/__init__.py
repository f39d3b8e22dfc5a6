"""Import alias: `open_seq2seq.*` -> `openseq2seq_amd.*`.

The reference's config files and user code import the plugin classes as
`from open_seq2seq.encoders import TDNNEncoder`, `from open_seq2seq.models import
Speech2Text`, ... (e.g. example_configs/speech2text/jasper10x5_LibriSpeech_nvgrad_masks.py:3-9).
With this directory on sys.path those imports resolve to the MI355X-native
implementations without touching the configs."""
import importlib
import importlib.abc
import importlib.util
import sys

_SRC, _DST = "open_seq2seq", "openseq2seq_amd"


class _AliasLoader(importlib.abc.Loader):
  def __init__(self, real_name):
    self.real_name = real_name

  def create_module(self, spec):
    return importlib.import_module(self.real_name)

  def exec_module(self, module):
    pass


class _AliasFinder(importlib.abc.MetaPathFinder):
  def find_spec(self, fullname, path, target=None):
    if fullname == _SRC or not fullname.startswith(_SRC + "."):
      return None
    real = _DST + fullname[len(_SRC):]
    try:
      if importlib.util.find_spec(real) is None:
        return None
    except (ImportError, ValueError):
      return None
    return importlib.util.spec_from_loader(fullname, _AliasLoader(real))


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
  sys.meta_path.insert(0, _AliasFinder())

import openseq2seq_amd as _impl  # noqa: E402

__path__ = list(_impl.__path__)
__version__ = _impl.__version__
