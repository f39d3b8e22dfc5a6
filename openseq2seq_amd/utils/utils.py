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
