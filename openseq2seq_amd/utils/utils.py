"""Configuration plumbing of the drop-in CLI: parameter-schema checks, nested dictionary overrides, the
`--a/b/c=value` command line and model construction.

What is a CONTRACT here — because user configs and scripts written against NVIDIA/OpenSeq2Seq depend on it — is
the behaviour of open_seq2seq/utils/utils.py: which keys are accepted, the wording of the ValueErrors
(`check_params`, :403-429), the "/"-joined names of nested overrides (`flatten_dict` / `nest_dict`, :296-349)
and the flags of `get_base_config` (:469-545). The code below is written against that behaviour, not copied
from it; tests/test_config_dropin.py runs the reference's own config files through it.
"""
from __future__ import print_function

import argparse
import ast
import contextlib
import copy
import runpy
import sys

_SCALARS = (int, float, str, bool)


def deco_print(line, offset=0, start="*** ", end='\n'):
  """The '*** '-prefixed progress lines of the reference's logs (scripts grep for them)."""
  sys.stdout.write("%s%s%s%s" % (start, " " * offset, line, end))
  sys.stdout.flush()


def _type_ok(value, want):
  """Does `value` satisfy one schema entry? None: anything; a list: one of its members; a type: an instance
  of it (text types interchangeable; an int where a float is asked for passes, as the reference's configs
  rely on)."""
  if want is None:
    return True, None
  if isinstance(want, list):
    return value in want, "{} has to be one of {}"
  if want is str:
    want = (str, type(u""))
  if want is float and isinstance(value, int) and not isinstance(value, bool):
    return True, None
  return isinstance(value, want), "{} has to be of type {}"


def check_params(config, required_dict, optional_dict):
  """Schema check of a plugin's parameter dictionary: every required key present, every present key known and
  of the declared kind. Raises ValueError with the reference's wording."""
  if required_dict is None or optional_dict is None:
    return
  # The ORDER in which problems are reported is the reference's (utils.py:407-429): the required table entry by
  # entry (absent, else of the wrong kind), then the kinds of the optional entries, unknown keys last — a config
  # with two mistakes raises the same message under both code bases.
  def kind(key, want):
    ok, msg = _type_ok(config[key], want)
    if not ok:
      shown = want if not (want is str) else (str, type(u""))
      raise ValueError(msg.format(key, shown))
  for key, want in required_dict.items():
    if key not in config:
      raise ValueError("{} parameter has to be specified".format(key))
    kind(key, want)
  for key, want in optional_dict.items():
    if key in config:
      kind(key, want)
  for key in config:
    if key not in required_dict and key not in optional_dict:
      raise ValueError("Unknown parameter: {}".format(key))


def nested_update(org_dict, upd_dict):
  """In-place merge of `upd_dict` into `org_dict`: sub-dictionaries merge key by key, anything else replaces.
  A dictionary arriving where the original holds a non-dictionary is an error."""
  stack = [(org_dict, upd_dict)]
  while stack:
    dst, src = stack.pop()
    for key, value in src.items():
      if isinstance(value, dict) and key in dst:
        if not isinstance(dst[key], dict):
          raise ValueError("Mismatch between org_dict and upd_dict at node {}".format(key))
        stack.append((dst[key], value))
      else:
        dst[key] = value


def flatten_dict(dct):
  """{'a': {'b': 1}, 'c': 2} -> {'a/b': 1, 'c': 2}: the scalar leaves (int / float / str / bool) of a nested
  dictionary under '/'-joined names — the names of the command-line overrides. Other leaves are dropped."""
  flat = {}

  def walk(prefix, node):
    for key, value in node.items():
      name = prefix + key
      if isinstance(value, dict):
        walk(name + '/', value)
      elif isinstance(value, _SCALARS):
        flat[name] = value
  walk('', dct)
  return flat


def nest_dict(flat_dict):
  """Inverse of flatten_dict."""
  root = {}
  for name, value in flat_dict.items():
    *parents, leaf = name.split('/')
    node = root
    for part in parents:
      node = node.setdefault(part, {})
    node[leaf] = value
  return root


@contextlib.contextmanager
def _tensorflow_token_module():
  """Make `import tensorflow as tf` inside a config file resolve to the token shim when
  the real TensorFlow is absent (the configs only use tf to name things)."""
  import importlib.util
  import sys
  installed = []
  if importlib.util.find_spec("tensorflow") is None and "tensorflow" not in sys.modules:
    from ..compat import tensorflow_shim as shim
    for name, mod in [("tensorflow", shim), ("tensorflow.nn", shim.nn),
                      ("tensorflow.contrib", shim.contrib), ("tensorflow.train", shim.train),
                      ("tensorflow.contrib.layers", shim.contrib.layers),
                      ("tensorflow.contrib.opt", shim.contrib.opt)]:
      sys.modules[name] = mod
      installed.append(name)
  try:
    yield sys.modules["tensorflow"]
  finally:
    for name in installed:
      sys.modules.pop(name, None)


def load_config_module(config_file):
  """runpy.run_path(config_file, init_globals={'tf': tf}) (utils.py:521) with the
  `open_seq2seq` alias package importable and the tf token module in place."""
  import os
  import sys
  repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  if repo not in sys.path:
    sys.path.insert(0, repo)
  import open_seq2seq  # noqa: F401  (installs the alias finder)
  with _tensorflow_token_module() as tf:
    # DL_REPLACE: the bare placeholder name text2speech/tacotron_gst.py expects the user to
    # substitute (dataset location); predefined so that the unedited config still loads and
    # falls back to synthetic batches when the files are absent
    return runpy.run_path(config_file, init_globals={
        'tf': tf, 'DL_REPLACE': os.environ.get('DL_REPLACE', '[REPLACE THIS TO THE PATH WITH YOUR DATA]')})


_MODES = ['train', 'eval', 'train_eval', 'infer', 'interactive_infer']
_FLAGS = ['continue_learning', 'no_dir_check', 'benchmark', 'enable_logs', 'use_xla_jit']


def get_base_config(args):
  """The reference's command line (utils.py:469-545): --config_file, --mode, --infer_output_file, the flags
  --continue_learning / --no_dir_check / --benchmark / --enable_logs / --use_xla_jit, --bench_steps,
  --bench_start, --debug_port, and one `--a/b/c=value` override per int / float / str / bool leaf of the
  config's base_params (typed like the value it replaces). Returns (args, base_config, base_model, module)."""
  cli = argparse.ArgumentParser(description='Experiment parameters')
  cli.add_argument("--config_file", required=True, help="Path to the configuration file")
  cli.add_argument("--mode", default='train', help='Could be "train", "eval", "train_eval" or "infer"')
  cli.add_argument("--infer_output_file", default='infer-out.txt')
  for flag in _FLAGS:
    cli.add_argument('--' + flag, dest=flag, action='store_true')
  cli.add_argument('--bench_steps', type=int, default='20')
  cli.add_argument('--bench_start', type=int)
  cli.add_argument('--debug_port', type=int)
  args, overrides = cli.parse_known_args(args)
  if args.mode not in _MODES:
    raise ValueError("Mode has to be one of %s" % (_MODES,))
  config_module = load_config_module(args.config_file)
  base_config = config_module.get('base_params', None)
  if base_config is None:
    raise ValueError('base_config dictionary has to be defined in the config file')
  base_model = config_module.get('base_model', None)
  if base_model is None:
    raise ValueError('base_config class has to be defined in the config file')
  leaves = argparse.ArgumentParser()
  for name, value in flatten_dict(base_config).items():
    # bool before int: a bool IS an int; its text form goes through literal_eval ("False" must not be truthy)
    if isinstance(value, bool):
      leaves.add_argument('--' + name, default=value, type=ast.literal_eval)
    else:
      leaves.add_argument('--' + name, default=value, type=type(value))
  nested_update(base_config, nest_dict(vars(leaves.parse_args(overrides))))
  return args, base_config, base_model, config_module


def create_model(args, base_config, config_module, base_model, hvd, device=None):
  """utils.py:791-864 (train / eval / infer model creation; --benchmark strips logging
  and sets max_steps = bench_steps, bench_start default 10)."""
  train_config = copy.deepcopy(base_config)
  eval_config = copy.deepcopy(base_config)
  infer_config = copy.deepcopy(base_config)
  if args.mode in ("train", "train_eval"):
    if 'train_params' in config_module:
      nested_update(train_config, copy.deepcopy(config_module['train_params']))
  if args.mode in ("eval", "train_eval"):
    if 'eval_params' in config_module:
      nested_update(eval_config, copy.deepcopy(config_module['eval_params']))
  if args.mode == "infer":
    if 'infer_params' in config_module:
      nested_update(infer_config, copy.deepcopy(config_module['infer_params']))
  if args.benchmark:
    deco_print("Adjusting config for benchmarking")
    for key in ['print_samples_steps', 'print_loss_steps', 'save_summaries_steps',
                'save_checkpoint_steps', 'logdir', 'eval_steps']:
      train_config.pop(key, None)
    train_config.pop('num_epochs', None)
    train_config['max_steps'] = args.bench_steps
    train_config['bench_start'] = args.bench_start if args.bench_start is not None else 10
  if args.mode in ("train", "train_eval"):
    model = base_model(params=train_config, mode="train", hvd=hvd, device=device)
    if args.mode == "train_eval":
      # utils.py:838-846: a second model in eval mode sharing the variables
      model.eval_model = base_model(params=eval_config, mode="eval", hvd=hvd, device=device)
      model.eval_model.compile()
  elif args.mode == "eval":
    model = base_model(params=eval_config, mode="eval", hvd=hvd, device=device)
  else:
    model = base_model(params=infer_config, mode="infer", hvd=hvd, device=device)
  model.compile()
  return model
