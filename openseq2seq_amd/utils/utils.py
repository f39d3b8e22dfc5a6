"""Config helpers with the reference's semantics (open_seq2seq/utils/utils.py)."""
from __future__ import print_function

import sys


def deco_print(line, offset=0, start="*** ", end='\n'):
  # utils.py:373-379
  print((start + " " * offset + line), end=end)
  sys.stdout.flush()


def check_params(config, required_dict, optional_dict):
  """utils.py:403-429 — unknown key => ValueError; wrong type => ValueError;
  a list in the schema enumerates the allowed values; None accepts anything."""
  if required_dict is None or optional_dict is None:
    return
  for pm, vals in required_dict.items():
    if pm not in config:
      raise ValueError("{} parameter has to be specified".format(pm))
    _check_one(pm, config[pm], vals)
  for pm, vals in optional_dict.items():
    if pm in config:
      _check_one(pm, config[pm], vals)
  for pm in config:
    if pm not in required_dict and pm not in optional_dict:
      raise ValueError("Unknown parameter: {}".format(pm))


def _check_one(pm, value, vals):
  if vals == str:
    vals = (str, type(u""))
  if vals is None:
    return
  if isinstance(vals, list):
    if value not in vals:
      raise ValueError("{} has to be one of {}".format(pm, vals))
    return
  if vals is float and isinstance(value, int) and not isinstance(value, bool):
    return  # ints are accepted where the reference's configs pass them for floats
  if not isinstance(value, vals):
    raise ValueError("{} has to be of type {}".format(pm, vals))


def nested_update(org_dict, upd_dict):
  # utils.py:351-363
  for key, value in upd_dict.items():
    if isinstance(value, dict):
      if key in org_dict:
        if not isinstance(org_dict[key], dict):
          raise ValueError("Mismatch between org_dict and upd_dict at node {}".format(key))
        nested_update(org_dict[key], value)
      else:
        org_dict[key] = value
    else:
      org_dict[key] = value


# ---------------------------------------------------------------------------
# config loading / CLI (open_seq2seq/utils/utils.py:296-349, 469-545, 791-864)
# ---------------------------------------------------------------------------
import argparse
import ast
import contextlib
import copy
import runpy


def flatten_dict(dct):
  flat_dict = {}
  for key, value in dct.items():
    if isinstance(value, (int, float, str, bool)):
      flat_dict.update({key: value})
    elif isinstance(value, dict):
      flat_dict.update({key + '/' + k: v for k, v in flatten_dict(dct[key]).items()})
  return flat_dict


def nest_dict(flat_dict):
  nst_dict = {}
  for key, value in flat_dict.items():
    nest_keys = key.split('/')
    cur_dict = nst_dict
    for i in range(len(nest_keys) - 1):
      if nest_keys[i] not in cur_dict:
        cur_dict[nest_keys[i]] = {}
      cur_dict = cur_dict[nest_keys[i]]
    cur_dict[nest_keys[-1]] = value
  return nst_dict


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


def get_base_config(args):
  """Same CLI as the reference (utils.py:469-545): --config_file, --mode, --benchmark,
  --bench_steps, --bench_start, --continue_learning, --no_dir_check, --enable_logs, plus
  `--a/b/c=value` overrides of any int/float/str/bool leaf of base_params."""
  parser = argparse.ArgumentParser(description='Experiment parameters')
  parser.add_argument("--config_file", required=True, help="Path to the configuration file")
  parser.add_argument("--mode", default='train',
                      help="Could be \"train\", \"eval\", \"train_eval\" or \"infer\"")
  parser.add_argument("--infer_output_file", default='infer-out.txt')
  parser.add_argument('--continue_learning', dest='continue_learning', action='store_true')
  parser.add_argument('--no_dir_check', dest='no_dir_check', action='store_true')
  parser.add_argument('--benchmark', dest='benchmark', action='store_true')
  parser.add_argument('--bench_steps', type=int, default='20')
  parser.add_argument('--bench_start', type=int)
  parser.add_argument('--debug_port', type=int)
  parser.add_argument('--enable_logs', dest='enable_logs', action='store_true')
  parser.add_argument('--use_xla_jit', dest='use_xla_jit', action='store_true')
  args, unknown = parser.parse_known_args(args)
  if args.mode not in ['train', 'eval', 'train_eval', 'infer', 'interactive_infer']:
    raise ValueError("Mode has to be one of ['train', 'eval', 'train_eval', 'infer', "
                     "'interactive_infer']")
  config_module = load_config_module(args.config_file)
  base_config = config_module.get('base_params', None)
  if base_config is None:
    raise ValueError('base_config dictionary has to be defined in the config file')
  base_model = config_module.get('base_model', None)
  if base_model is None:
    raise ValueError('base_config class has to be defined in the config file')
  parser_unk = argparse.ArgumentParser()
  for pm, value in flatten_dict(base_config).items():
    if type(value) == int or type(value) == float or isinstance(value, str):
      parser_unk.add_argument('--' + pm, default=value, type=type(value))
    elif type(value) == bool:
      parser_unk.add_argument('--' + pm, default=value, type=ast.literal_eval)
  config_update = parser_unk.parse_args(unknown)
  nested_update(base_config, nest_dict(vars(config_update)))
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
