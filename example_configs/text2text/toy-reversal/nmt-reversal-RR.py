# pylint: skip-file
"""The reference's RNN acceptance configuration for the toy reversal task
(example_configs/text2text/toy-reversal/nmt-reversal-RR.py): 1-layer bidirectional LSTM-128
encoder, 1-layer LSTM-128 decoder with Luong (multiplicative) attention, Adam 1e-3 with
gradient-norm clipping at 3, BasicSequenceLoss; evaluation decodes greedily. Same values as the
reference file; the evaluation data layer does not repeat (this engine evaluates one pass)."""
from __future__ import absolute_import, division, print_function
import tensorflow as tf

from open_seq2seq.models import Text2Text
from open_seq2seq.encoders import BidirectionalRNNEncoderWithEmbedding
from open_seq2seq.decoders import RNNDecoderWithAttention, BeamSearchRNNDecoderWithAttention
from open_seq2seq.data.text2text.text2text import ParallelTextDataLayer
from open_seq2seq.losses import BasicSequenceLoss
from open_seq2seq.data.text2text.text2text import SpecialTextTokens
from open_seq2seq.optimizers.lr_policies import fixed_lr

data_root = "toy_text_data/"
base_model = Text2Text

decoder_arch = {
  "core_cell": tf.nn.rnn_cell.LSTMCell,
  "core_cell_params": {"num_units": 128},
  "decoder_layers": 1,
  "decoder_dp_input_keep_prob": 0.8,
  "decoder_dp_output_keep_prob": 1.0,
  "decoder_use_skip_connections": False,
  "GO_SYMBOL": SpecialTextTokens.S_ID.value,
  "END_SYMBOL": SpecialTextTokens.EOS_ID.value,
  "tgt_emb_size": 128,
  "attention_type": "luong",
  "luong_scale": False,
  "attention_layer_size": 128,
}

base_params = {
  "use_horovod": False,
  "num_gpus": 1,
  "batch_size_per_gpu": 64,
  "max_steps": 800,
  "print_loss_steps": 100,
  "eval_steps": 400,
  "save_checkpoint_steps": 300,
  "logdir": "ReversalTask-RNN-RNN",
  "optimizer": "Adam",
  "optimizer_params": {"epsilon": 1e-4},
  "lr_policy": fixed_lr,
  "lr_policy_params": {"learning_rate": 0.001},
  "max_grad_norm": 3.0,
  "dtype": tf.float32,

  "encoder": BidirectionalRNNEncoderWithEmbedding,
  "encoder_params": {
    "core_cell": tf.nn.rnn_cell.LSTMCell,
    "core_cell_params": {"num_units": 128, "forget_bias": 1.0},
    "encoder_layers": 1,
    "encoder_dp_input_keep_prob": 0.8,
    "encoder_dp_output_keep_prob": 1.0,
    "encoder_use_skip_connections": False,
    "src_emb_size": 128,
  },

  "decoder": RNNDecoderWithAttention,
  "decoder_params": decoder_arch,

  "loss": BasicSequenceLoss,
  "loss_params": {"offset_target_by_one": True, "average_across_timestep": False, "do_mask": True},
}


def _data(split, **kw):
  d = {
    "src_vocab_file": data_root + "vocab/source.txt",
    "tgt_vocab_file": data_root + "vocab/target.txt",
    "source_file": data_root + split + "/source.txt",
    "target_file": data_root + split + "/target.txt",
    "max_length": 56, "delimiter": " ", "special_tokens_already_in_vocab": False,
  }
  d.update(kw)
  return d


train_params = {"data_layer": ParallelTextDataLayer, "data_layer_params": _data("train", shuffle=True, repeat=True)}
eval_params = {"data_layer": ParallelTextDataLayer, "data_layer_params": _data("dev", shuffle=False, repeat=False)}
infer_params = {
  "batch_size_per_gpu": 8,
  "decoder": BeamSearchRNNDecoderWithAttention,
  "decoder_params": dict(decoder_arch, beam_width=5, length_penalty=1.0,
                         PAD_SYMBOL=SpecialTextTokens.PAD_ID.value),
  "data_layer": ParallelTextDataLayer,
  "data_layer_params": _data("test", shuffle=False, repeat=False, max_length=256),
}
