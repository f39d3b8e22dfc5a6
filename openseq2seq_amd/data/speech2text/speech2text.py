"""Speech2TextDataLayer — parameter schema and batch contract of
open_seq2seq/data/speech2text/speech2text.py:22-485.

Batch contract (what build_graph()/input_tensors deliver to the model):
  source_tensors = [features [B, T, num_audio_features], src_length int32 [B]]
      T = max frames in the batch padded up to a multiple of `pad_to`
      (speech2text.py:313-317), frames = 1 + num_samples // hop (librosa backend)
  target_tensors = [transcript ids int32 [B, Lmax] (pad value 0), tgt_length int32 [B]]
      vocabulary = characters of vocab_file; tgt_vocab_size = len(vocab) + 1 (CTC blank
      is the LAST id, speech2text.py:123-125).
The tf.data/py_func plumbing of the reference is host-side and not re-created; the
feature math (speech_utils.get_speech_features_librosa) runs on the GPU
(csrc/logmel.hip) — see data/speech2text/speech_utils.py. `synthetic_batch` draws
the synthetic workload SURVEY.md §8d defines for benchmarking.
"""
from __future__ import absolute_import, division, print_function

import csv
import os
import queue
import threading

import numpy as np
import torch

from ..data_layer import DataLayer

DEFAULT_VOCAB = [" "] + [chr(ord("a") + i) for i in range(26)] + ["'"]


def load_pre_existing_vocabulary(path, read_chars=True):
  """open_seq2seq/data/utils.py: one token per line; first char when read_chars."""
  idx = 0
  vocab = {}
  with open(path, "r", encoding="utf-8") as f:
    for line in f:
      if len(line.rstrip("\n")) == 0 and not line.startswith(" "):
        continue
      tok = line[0] if read_chars else line.rstrip("\n").split("\t")[0]
      vocab[tok] = idx
      idx += 1
  return vocab


class Speech2TextDataLayer(DataLayer):
  @staticmethod
  def get_required_params():
    return dict(DataLayer.get_required_params(), **{
        'num_audio_features': int,
        'input_type': ['spectrogram', 'mfcc', 'logfbank'],
        'vocab_file': None,    # str; None selects the 28-character toy vocabulary
        'dataset_files': list,
    })

  @staticmethod
  def get_optional_params():
    return dict(DataLayer.get_optional_params(), **{
        'backend': ['psf', 'librosa'], 'augmentation': dict, 'pad_to': int,
        'max_duration': float, 'min_duration': float, 'bpe': bool, 'autoregressive': bool,
        'syn_enable': bool, 'syn_subdirs': list, 'window_size': float,
        'window_stride': float, 'dither': float, 'norm_per_feature': bool,
        'window': ['hanning', 'hamming', 'none'], 'num_fft': int,
        'precompute_mel_basis': bool, 'sample_freq': int, 'gain': float,
        'features_mean': np.ndarray, 'features_std_dev': np.ndarray,
    })

  def __init__(self, params, model, num_workers, worker_id):
    super(Speech2TextDataLayer, self).__init__(params, model, num_workers, worker_id)
    p = self.params
    if p.get('bpe', False) or p.get('autoregressive', False):
      raise NotImplementedError("bpe / autoregressive targets")
    if p['vocab_file']:
      p['char2idx'] = load_pre_existing_vocabulary(p['vocab_file'], read_chars=True)
    else:
      p['char2idx'] = {c: i for i, c in enumerate(DEFAULT_VOCAB)}
    p['idx2char'] = {i: c for c, i in p['char2idx'].items()}
    # add one for the implied blank token (speech2text.py:123-125)
    p['tgt_vocab_size'] = len(p['char2idx']) + 1
    self._input_tensors = {}
    p.setdefault('min_duration', -1.0)
    p.setdefault('max_duration', -1.0)
    p.setdefault('window_size', 20e-3)
    p.setdefault('window_stride', 10e-3)
    p.setdefault('sample_freq', 16000)
    aug = p.get('augmentation') or {}
    if 'n_freq_mask' in aug and aug.get('width_freq_mask', 10) > p['num_audio_features']:
      raise ValueError("'width_freq_mask'={} should be smaller than 'num_audio_features'={}".format(
          aug.get('width_freq_mask', 10), p['num_audio_features']))
    if 'time_stretch_ratio' in aug:      # speech2text.py:194-196
      aug['speed_perturbation_ratio'] = aug['time_stretch_ratio']
    # ---- dataset files (speech2text.py:139-153): csv with wav_filename[, wav_filesize], transcript
    self.all_files = None
    self._files = None
    self._front = None
    if not p.get('interactive', False) and self._csvs_present():
      rows = []
      for path in p['dataset_files']:
        base = os.path.dirname(os.path.abspath(path))
        with open(path, "r", encoding="utf-8", newline="") as f:
          for r in csv.DictReader(f):
            wav = r['wav_filename']
            if not os.path.isabs(wav) and not os.path.exists(wav) and os.path.exists(os.path.join(base, wav)):
              wav = os.path.join(base, wav)       # convenience: paths relative to the csv
            rows.append((wav, r.get('transcript', '')))
      self.all_files = rows
      self._files = self.split_data(rows)

  def _csvs_present(self):
    files = self.params.get('dataset_files') or []
    return len(files) > 0 and all(isinstance(f, str) and os.path.exists(f) for f in files)

  def has_files(self):
    return self._files is not None and len(self._files) > 0

  def split_data(self, data):
    """speech2text.py:198-208: eval / infer data is sharded over the workers, training data
    is not (every worker shuffles with its own seed)."""
    if self.params['mode'] != 'train' and self._num_workers is not None:
      size = len(data)
      start = size // self._num_workers * self._worker_id
      end = size if self._worker_id == self._num_workers - 1 else size // self._num_workers * (self._worker_id + 1)
      return data[start:end]
    return data

  def build_graph(self):
    return self

  @property
  def input_tensors(self):
    return self._input_tensors

  def get_size_in_samples(self):
    return len(self._files) if self._files is not None else None

  # ---- real files -> batches ---------------------------------------------------------------------
  def encode_transcript(self, transcript):
    """speech2text.py:407-414 (character targets)."""
    c2i = self.params['char2idx']
    return np.array([c2i[c] for c in transcript], np.int32)

  def _load_example(self, wav, transcript, rng):
    """Host part of _parse_audio_transcript_element (:395-431): decode the file, check the
    sample rate, draw this sample's augmentation parameters (in the reference's order:
    stretch, then noise level, then SpecAugment boxes)."""
    from .speech_utils import read_wav
    p = self.params
    sr, signal = read_wav(wav)
    if sr != p['sample_freq']:
      raise ValueError(("The sampling frequency set in params {} does not match the "
                        "frequency {} read from file {}").format(p['sample_freq'], sr, wav))
    if signal.dtype != np.int16:
      signal = signal.astype(np.float32)
    aug = p.get('augmentation') or {}
    ratio, amp = 1.0, 0.0
    if aug:      # applied in whichever mode's data_layer_params carry it (speech_utils.py:355-356)
      if 'speed_perturbation_ratio' in aug:       # speech_utils.py:253-268
        spr = aug['speed_perturbation_ratio']
        stretch = -1
        if isinstance(spr, list):
          stretch = rng.choice(spr)
        elif spr > 0:
          stretch = 1.0 + (2.0 * rng.rand() - 1.0) * spr
        if stretch > 0:
          ratio = float(int(sr * stretch)) / float(sr)     # resampy: sr_new / sr_orig
      if 'noise_level_min' in aug and 'noise_level_max' in aug:
        amp = 10.0 ** (rng.randint(low=aug['noise_level_min'], high=aug['noise_level_max']) / 20.0)
    n_out = int(len(signal) * ratio) if ratio != 1.0 else len(signal)
    return dict(signal=signal, n_out=n_out, ratio=ratio, amp=amp,
                target=self.encode_transcript(transcript) if transcript is not None else None,
                duration=n_out * 1.0 / sr)

  def _spec_masks(self, frames, rng):
    """SpecAugment boxes of one sample (speech_utils.py:419-433) as (t0, t1, f0, f1)."""
    aug = self.params.get('augmentation') or {}
    F = self.params['num_audio_features']
    boxes = []
    for _ in range(aug.get('n_freq_mask', 0)):
      band = rng.randint(aug.get('width_freq_mask', 10) + 1)
      base = rng.randint(0, F - band)
      boxes.append((0, frames, base, base + band))
    for _ in range(aug.get('n_time_mask', 0)):
      band = rng.randint(aug.get('width_time_mask', 50) + 1)
      if frames - band > 0:
        base = rng.randint(frames - band)
        boxes.append((base, base + band, 0, F))
      else:
        boxes.append((0, 0, 0, 0))
    return boxes

  def _host_batches(self, seed, drop_remainder):
    """Generator of host-side batches (lists of examples): shuffle -> decode -> duration filter
    (:230-239) -> groups of batch_size, like the tf.data pipeline of build_graph (:217-262)."""
    p = self.params
    B = p['batch_size']
    rng = np.random.RandomState(seed)
    files = self._files
    while True:
      order = rng.permutation(len(files)) if p.get('shuffle', False) else np.arange(len(files))
      cur = []
      for j in order:
        wav, txt = files[j]
        ex = self._load_example(wav, txt if p['mode'] != 'infer' else None, rng)
        ex['index'] = int(j)
        if p['max_duration'] > 0 and ex['duration'] > p['max_duration']:
          continue
        if p['min_duration'] > 0 and ex['duration'] < p['min_duration']:
          continue
        ex['masks'] = self._spec_masks(self.frames_for_samples(ex['n_out']), rng)
        cur.append(ex)
        if len(cur) == B:
          yield cur
          cur = []
      if cur and not drop_remainder:
        yield cur
      if not p.get('repeat', False):
        return

  def _device_batch(self, exs, device, seed):
    """Pad + upload one host batch and run the GPU front end: normalise / speed-perturb /
    noise (os2s_augment_signal) when any sample asks for it, log-mel features (os2s_logmel),
    SpecAugment boxes (os2s_spec_augment)."""
    from ... import capi
    from .speech_utils import KAISER_BEST, make_front_end, sinc_window
    p = self.params
    if self._front is None or self._front.device != device:
      self._front = make_front_end(p, device)
      win, self._num_table = sinc_window(**KAISER_BEST)
      self._interp_win = torch.from_numpy(win).to(device)
    B = len(exs)
    n_in = np.array([len(e['signal']) for e in exs], np.int32)
    n_out = np.array([e['n_out'] for e in exs], np.int32)
    all_i16 = all(e['signal'].dtype == np.int16 for e in exs)
    sig = np.zeros((B, int(n_in.max())), np.int16 if all_i16 else np.float32)
    for b, e in enumerate(exs):
      sig[b, :n_in[b]] = e['signal']
    sig = torch.from_numpy(sig).to(device, non_blocking=True)
    if any(e['ratio'] != 1.0 or e['amp'] > 0 for e in exs):
      sig = capi.augment_signal(
          sig, torch.from_numpy(n_in).to(device), torch.from_numpy(n_out).to(device),
          torch.tensor([e['ratio'] for e in exs], dtype=torch.float64, device=device),
          torch.tensor([e['amp'] for e in exs], dtype=torch.float32, device=device),
          self._interp_win, self._num_table, int(n_out.max()), seed=seed,
          # psf path: augmentation runs on the raw samples, the feature kernel normalises
          # (speech_utils.py:469-473); librosa path: normalise first (:334-356)
          fixed_gain=1.0 if self._psf() else (p.get('gain') if p.get('gain') is not None else -1.0))
      feats, frames, _ = self._front_call(sig, n_out, seed, fixed_gain=1.0)
    else:
      feats, frames, _ = self._front_call(sig, n_out, seed)
    nmask = max((len(e['masks']) for e in exs), default=0)
    if nmask:
      mk = np.zeros((B, nmask, 4), np.int32)
      for b, e in enumerate(exs):
        for m, box in enumerate(e['masks']):
          mk[b, m] = box
      capi.spec_augment(feats, torch.from_numpy(mk).to(device))
    batch = {'source_tensors': [feats, frames], 'num_frames': int(sum(self.frames_for_samples(n) for n in n_out)),
             'padded_frames': int(B * feats.shape[1]),
             'source_ids': torch.tensor([e['index'] for e in exs], dtype=torch.int32, device=device)}
    if not self._psf():
      # the frame counts as the host computes them (= `frames`, tests/test_speech_data_gpu.py): a hint for the
      # convolution launcher, see TDNNEncoder._encode
      batch['source_lengths_host'] = np.array([self.frames_for_samples(n) for n in n_out], np.int32)
    if exs[0]['target'] is not None:
      tl = np.array([len(e['target']) for e in exs], np.int32)
      tgt = np.zeros((B, max(int(tl.max()), 1)), np.int32)       # target_pad_value = 0 (:136)
      for b, e in enumerate(exs):
        tgt[b, :tl[b]] = e['target']
      batch['target_tensors'] = [torch.from_numpy(tgt).to(device), torch.from_numpy(tl).to(device)]
    return batch

  def _front_call(self, sig, n_out, seed, fixed_gain=None):
    fe = self._front
    if fixed_gain is not None:
      saved, fe.gain = fe.gain, fixed_gain
      try:
        return fe(sig, torch.from_numpy(n_out).to(sig.device), max_samples=int(n_out.max()), seed=seed)
      finally:
        fe.gain = saved
    return fe(sig, torch.from_numpy(n_out).to(sig.device), max_samples=int(n_out.max()), seed=seed)

  def iterate_batches(self, device, seed=0, drop_remainder=None, prefetch=4):
    """Yields batch dicts (forever when repeat=True). File decoding and the augmentation draws
    run in a background thread `prefetch` batches ahead (the reference's tf.data prefetch +
    parallel py_func map); the feature kernels run on the caller's stream."""
    if not self.has_files():
      raise ValueError("no dataset files to iterate (dataset_files missing on disk)")
    if drop_remainder is None:
      drop_remainder = self.params['mode'] == 'train'
    q = queue.Queue(maxsize=max(int(prefetch), 1))
    stop = threading.Event()
    done = object()

    def worker():
      try:
        for hb in self._host_batches(seed, drop_remainder):
          while not stop.is_set():
            try:
              q.put(hb, timeout=0.1)
              break
            except queue.Full:
              continue
          if stop.is_set():
            return
        q.put(done)
      except BaseException as e:      # surface loader errors in the consumer
        q.put(e)

    th = threading.Thread(target=worker, daemon=True)
    th.start()
    step = 0
    try:
      while True:
        item = q.get()
        if item is done:
          return
        if isinstance(item, BaseException):
          raise item
        yield self._device_batch(item, device, seed * 1000003 + step)
        step += 1
    finally:
      stop.set()

  # ------------------------------------------------------------------------
  def _psf(self):
    return self.params.get('backend', 'psf') == 'psf' and self.params.get('input_type') == 'spectrogram'

  def frames_for_samples(self, n_samples):
    sr = self.params.get('sample_freq', 16000)
    hop = int(sr * self.params.get('window_stride', 10e-3))
    if self._psf():     # framesig: 1 + ceil((n - win) / hop), before the pad_to rounding
      win = int(sr * self.params.get('window_size', 20e-3))
      return 1 if n_samples <= win else 1 + -(-(n_samples - win) // hop)
    return 1 + n_samples // hop

  def synthetic_batch(self, device, seed, fixed_frames=None, min_dur=2.0):
    """Synthetic training batch: durations U[min_dur, max_duration] s,
    features N(0,1) in bf16 (what per-feature normalisation produces), T padded to
    `pad_to`; labels U{0..V-2} of length U[10, T'/2] where T' = frames after the
    stride-2 encoder layer. Returns the batch dict for Model.train_step."""
    p = self.params
    B = p['batch_size']
    rng = np.random.RandomState(seed)
    sr = p.get('sample_freq', 16000)
    if fixed_frames is None:
      dur = rng.uniform(min_dur, p.get('max_duration', 16.7), size=B)
      frames = np.array([self.frames_for_samples(int(d * sr)) for d in dur], np.int32)
    else:
      frames = np.full(B, fixed_frames, np.int32)
    pad_to = p.get('pad_to', 8)
    T = int(-(-frames.max() // pad_to) * pad_to)
    F = p['num_audio_features']
    g = torch.Generator(device="cpu").manual_seed(seed)
    feats = torch.randn(B, T, F, generator=g).to(torch.bfloat16)
    V = p['tgt_vocab_size']
    tlen = np.array([rng.randint(10, max(11, f // 4)) for f in frames], np.int32)
    Lmax = int(tlen.max())
    tgt = rng.randint(0, V - 1, size=(B, Lmax)).astype(np.int32)
    for b in range(B):
      tgt[b, tlen[b]:] = 0
      feats[b, frames[b]:] = 0
    return {
        'source_tensors': [feats.to(device), torch.from_numpy(frames).to(device)],
        'source_lengths_host': frames.astype(np.int32).copy(),      # the same lengths, kept on the host
        'target_tensors': [torch.from_numpy(tgt).to(device), torch.from_numpy(tlen).to(device)],
        'num_frames': int(frames.sum()), 'padded_frames': int(B * T),
    }
