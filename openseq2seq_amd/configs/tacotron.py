"""Programmatic equivalent of example_configs/text2speech/tacotron_gst.py (Tacotron 2 with
global style tokens on M-AILABS: output_type "both" = 80 mel + 401 magnitude bins, batch 32
per GPU, Adam + exp_decay, max_grad_norm 1, L2 1e-6)."""
from ..data.text2speech.text2speech import Text2SpeechDataLayer
from ..decoders.tacotron2_decoder import Tacotron2Decoder
from ..encoders.tacotron2_encoder import Tacotron2Encoder
from ..losses.text2speech_loss import Text2SpeechLoss
from ..models.text2speech import Text2SpeechTacotron
from ..optimizers.lr_policies import exp_decay


def _conv(k, c, act=None, two_d=False):
  d = {"kernel_size": [k, k] if two_d else [k], "stride": [2, 2] if two_d else [1],
       "num_channels": c, "padding": "SAME"}
  if act is not False:
    d["activation_fn"] = act
  return d


def tacotron_gst_config(batch_size_per_gpu=32, max_steps=100000, style=True, dtype="mixed",
                        fp8_weights=False):
  enc = {
      "cnn_dropout_prob": 0.5, "rnn_dropout_prob": 0., "src_emb_size": 512,
      "conv_layers": [_conv(5, 512, False)] * 3, "activation_fn": "relu",
      "num_rnn_layers": 1, "rnn_cell_dim": 256, "rnn_unidirectional": False,
      "use_cudnn_rnn": True, "rnn_type": "CudnnLSTM", "zoneout_prob": 0.,
      "data_format": "channels_last",
  }
  if style:
    enc["style_embedding_enable"] = True
    enc["style_embedding_params"] = {
        "conv_layers": [_conv(3, c, False, True) for c in (32, 32, 64, 64, 128, 128)],
        "num_rnn_layers": 1, "rnn_cell_dim": 128, "rnn_unidirectional": True,
        "rnn_type": "GRUCell", "emb_size": 512, "attention_layer_size": 512,
        "num_tokens": 32, "num_heads": 8,
    }
  base_params = {
      "random_seed": 0, "use_horovod": True, "batch_size_per_gpu": batch_size_per_gpu,
      "os2s_side_stream": False,     # engine knob (models/model.py): 128.9 vs 131.9 ms/step on this model
      "max_steps": max_steps, "max_grad_norm": 1.,
      "optimizer": "Adam", "optimizer_params": {},
      "lr_policy": exp_decay,
      "lr_policy_params": {"learning_rate": 1e-3, "decay_steps": 10000, "decay_rate": 0.1,
                           "use_staircase_decay": False, "begin_decay_at": 20000, "min_lr": 1e-5},
      "dtype": dtype, "loss_scaling": "Backoff",
      "regularizer": "l2_regularizer", "regularizer_params": {"scale": 1e-6},
      "encoder": Tacotron2Encoder, "encoder_params": enc,
      "decoder": Tacotron2Decoder,
      "decoder_params": {
          "zoneout_prob": 0., "dropout_prob": 0.1, "attention_type": "location",
          "attention_layer_size": 128, "attention_bias": True, "decoder_cell_units": 1024,
          "decoder_cell_type": "LSTMCell", "decoder_layers": 2, "enable_prenet": True,
          "prenet_layers": 2, "prenet_units": 256, "enable_postnet": True,
          "postnet_keep_dropout_prob": 0.5, "postnet_data_format": "channels_last",
          "postnet_conv_layers": [_conv(5, 512, "tanh")] * 4 + [_conv(5, -1, None)],
          "mask_decoder_sequence": True, "parallel_iterations": 32,
          # BASELINE.json configs[4] "fp8 weights": e4m3 copies of the decoder LSTM stack's recurrent
          # weights (our extension; the reference has no fp8)
          "fp8_weights": bool(fp8_weights),
      },
      "loss": Text2SpeechLoss, "loss_params": {"use_mask": True},
      "data_layer": Text2SpeechDataLayer,
      "data_layer_params": {
          "dataset": "MAILABS", "num_audio_features": {"mel": 80, "magnitude": 401},
          "output_type": "both", "vocab_file": "open_seq2seq/test_utils/vocab_tts.txt",
          "dataset_location": "", "dataset_files": [], "mag_power": 1, "pad_EOS": True,
          "feature_normalize": False, "feature_normalize_mean": 0., "feature_normalize_std": 1.,
          "data_min": {"mel": 1e-2, "magnitude": 1e-5}, "mel_type": "htk", "trim": True,
          "duration_max": 1024, "duration_min": 24, "exp_mag": True, "shuffle": True,
          "style_input": "wav" if style else None,
      },
  }
  return Text2SpeechTacotron, base_params
