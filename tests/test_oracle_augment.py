"""Resampling oracle (oracle/augment.py, restating resampy's published algorithm) cross-checked
against an independent polyphase resampler on band-limited signals, and the host-side logic of
the real-file Speech2TextDataLayer (csv parsing, sharding, duration filter, targets, SpecAugment
draws) on generated wav files."""
import csv
import os

import numpy as np
import pytest
from scipy import signal as sps
from scipy.io import wavfile

from oracle import augment as oa


def _tone_mix(n, sr, rng):
  t = np.arange(n) / sr
  x = np.zeros(n)
  for f in (220.0, 1333.0, 3100.0):
    x += rng.rand() * np.sin(2 * np.pi * f * t + rng.rand())
  return (x / np.abs(x).max()).astype(np.float32)


@pytest.mark.parametrize("sr_new", [17600, 14400])
def test_resample_matches_polyphase(sr_new):
  sr = 16000
  rng = np.random.RandomState(0)
  x = _tone_mix(4000, sr, rng)
  y = oa.resample(x, sr, sr_new)
  assert y.shape[0] == int(4000 * sr_new / sr)
  g = np.gcd(sr, sr_new)
  ref = sps.resample_poly(x.astype(np.float64), sr_new // g, sr // g, window=("kaiser", 14.0))
  m = slice(300, len(y) - 300)                  # away from the edges (different edge handling)
  err = np.abs(y[m] - ref[:len(y)][m]).max()
  assert err < 5e-3, err


def test_resample_identity_ratio_close():
  rng = np.random.RandomState(1)
  x = _tone_mix(2000, 16000, rng)
  y = oa.resample(x, 16000, 16000)
  assert np.abs(y[200:-200] - x[200:-200]).max() < 2e-3


def _make_dataset(tmp, n=7, sr=16000):
  rng = np.random.RandomState(3)
  rows = []
  for i in range(n):
    dur = 0.4 + 0.2 * i
    x = (_tone_mix(int(dur * sr), sr, rng) * 20000).astype(np.int16)
    path = os.path.join(tmp, "u%d.wav" % i)
    wavfile.write(path, sr, x)
    rows.append((path, os.path.getsize(path), "ab c'" + "d" * i))
  csv_path = os.path.join(tmp, "data.csv")
  with open(csv_path, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["wav_filename", "wav_filesize", "transcript"])
    w.writerows(rows)
  return csv_path, rows


def _layer(csv_path, mode="train", **kw):
  from openseq2seq_amd.data.speech2text.speech2text import Speech2TextDataLayer
  params = dict(mode=mode, batch_size=2, num_audio_features=64, input_type="logfbank", vocab_file=None,
                dataset_files=[csv_path], backend="librosa", shuffle=False, repeat=False)
  workers = kw.pop("workers", (1, 0))
  params.update(kw)
  return Speech2TextDataLayer(params, None, workers[0], workers[1])


def test_data_layer_host_logic(tmp_path):
  csv_path, rows = _make_dataset(str(tmp_path))
  dl = _layer(csv_path, max_duration=1.45, min_duration=0.5)
  assert dl.has_files() and dl.get_size_in_samples() == 7
  assert dl.params['tgt_vocab_size'] == 29
  batches = list(dl._host_batches(seed=0, drop_remainder=True))
  # durations 0.4 .. 1.6: min 0.5 drops u0, max 1.45 drops u6 -> u1..u5 -> 2 full batches of 2
  assert [[e['index'] for e in b] for b in batches] == [[1, 2], [3, 4]]
  e = batches[0][0]
  assert e['target'].tolist() == [1, 2, 0, 3, 27, 4] and e['ratio'] == 1.0 and e['amp'] == 0.0
  assert e['signal'].dtype == np.int16 and len(e['signal']) == int(0.6 * 16000)
  assert len(list(dl._host_batches(0, drop_remainder=False))) == 3
  # eval data is sharded over workers, training data is not (speech2text.py:198-208)
  assert _layer(csv_path, mode="eval", workers=(2, 1)).get_size_in_samples() == 4
  assert _layer(csv_path, mode="train", workers=(2, 1)).get_size_in_samples() == 7
  with pytest.raises(ValueError):
    list(_layer(csv_path, sample_freq=8000)._host_batches(0, True))


def test_augmentation_draws(tmp_path):
  csv_path, _ = _make_dataset(str(tmp_path), n=4)
  aug = dict(speed_perturbation_ratio=0.1, noise_level_min=-90, noise_level_max=-60, n_freq_mask=2,
             n_time_mask=2, width_freq_mask=6, width_time_mask=10)
  dl = _layer(csv_path, augmentation=aug)
  for b in dl._host_batches(seed=5, drop_remainder=False):
    for e in b:
      assert 0.9 <= e['ratio'] <= 1.1 and e['n_out'] == int(len(e['signal']) * e['ratio'])
      assert 10 ** (-90 / 20.) <= e['amp'] < 10 ** (-60 / 20.)
      frames = dl.frames_for_samples(e['n_out'])
      assert len(e['masks']) == 4
      for (t0, t1, f0, f1) in e['masks'][:2]:
        assert (t0, t1) == (0, frames) and 0 <= f0 <= f1 <= 64 and f1 - f0 <= 6
      for (t0, t1, f0, f1) in e['masks'][2:]:
        assert (f0, f1) == (0, 64) and 0 <= t0 <= t1 <= frames and t1 - t0 <= 10
  dl2 = _layer(csv_path, augmentation=dict(speed_perturbation_ratio=[0.9, 1.1]))
  ratios = {e['ratio'] for b in dl2._host_batches(1, False) for e in b}
  assert ratios <= {14400 / 16000., 17600 / 16000.}
  with pytest.raises(ValueError):
    _layer(csv_path, augmentation=dict(n_freq_mask=1, width_freq_mask=100))
