# pylint: skip-file
"""en-de-nmt-small architecture (example_configs/text2text/en-de/en-de-nmt-small.py in the
reference: 2-layer bidirectional LSTM-512 encoder, 2-layer GNMT-v2 attention decoder,
BasicSequenceLoss, Adam 1e-3 + LARC) on the toy reversal corpus
(python -m openseq2seq_amd.test_utils.create_reversed_examples), i.e. BASELINE.json's
configs[0]: data paths + special_tokens_already_in_vocab False as in
toy-reversal/nmt-reversal-RR.py."""
from __future__ import absolute_import, division, print_function
import tensorflow as tf

from open_seq2seq.models import Text2Text
from open_seq2seq.encoders import BidirectionalRNNEncoderWithEmbedding
from open_seq2seq.decoders import RNNDecoderWithAttention
from open_seq2seq.data.text2text.text2text import ParallelTextDataLayer
from open_seq2seq.losses import BasicSequenceLoss
from open_seq2seq.data.text2text.text2text import SpecialTextTokens
from open_seq2seq.optimizers.lr_policies import fixed_lr

data_root = "toy_text_data/"

base_model = Text2Text

base_params = {
  "use_horovod": False,
  "num_gpus": 1,
  "max_steps": 800,
  "batch_size_per_gpu": 128,
  "print_loss_steps": 50,
  "eval_steps": 400,
  "logdir": "nmt-small-reversal",
  "optimizer": "Adam",
  "optimizer_params": {},
  "lr_policy": fixed_lr,
  "lr_policy_params": {"learning_rate": 0.001},
  "larc_params": {"larc_eta": 0.001},
  "dtype": "mixed",
  "loss_scaling": "Backoff",

  "encoder": BidirectionalRNNEncoderWithEmbedding,
  "encoder_params": {
    "initializer": tf.glorot_uniform_initializer,
    "core_cell": tf.nn.rnn_cell.LSTMCell,
    "core_cell_params": {"num_units": 512, "forget_bias": 1.0},
    "encoder_layers": 2,
    "encoder_dp_input_keep_prob": 0.8,
    "encoder_dp_output_keep_prob": 1.0,
    "encoder_use_skip_connections": False,
    "src_emb_size": 512,
    "use_swap_memory": True,
  },

  "decoder": RNNDecoderWithAttention,
  "decoder_params": {
    "initializer": tf.glorot_uniform_initializer,
    "core_cell": tf.nn.rnn_cell.LSTMCell,
    "core_cell_params": {"num_units": 512, "forget_bias": 1.0},
    "decoder_layers": 2,
    "decoder_dp_input_keep_prob": 0.8,
    "decoder_dp_output_keep_prob": 1.0,
    "decoder_use_skip_connections": False,
    "GO_SYMBOL": SpecialTextTokens.S_ID.value,
    "END_SYMBOL": SpecialTextTokens.EOS_ID.value,
    "tgt_emb_size": 512,
    "attention_type": "gnmt_v2",
    "attention_layer_size": 512,
    "use_swap_memory": True,
  },

  "loss": BasicSequenceLoss,
  "loss_params": {
    "offset_target_by_one": True,
    "average_across_timestep": False,
    "do_mask": True,
  },
}

train_params = {
  "data_layer": ParallelTextDataLayer,
  "data_layer_params": {
    "src_vocab_file": data_root + "vocab/source.txt",
    "tgt_vocab_file": data_root + "vocab/target.txt",
    "source_file": data_root + "train/source.txt",
    "target_file": data_root + "train/target.txt",
    "delimiter": " ",
    "shuffle": True,
    "repeat": True,
    "max_length": 56,
    "special_tokens_already_in_vocab": False,
  },
}

eval_params = {
  "batch_size_per_gpu": 128,
  "data_layer": ParallelTextDataLayer,
  "data_layer_params": {
    "src_vocab_file": data_root + "vocab/source.txt",
    "tgt_vocab_file": data_root + "vocab/target.txt",
    "source_file": data_root + "dev/source.txt",
    "target_file": data_root + "dev/target.txt",
    "delimiter": " ",
    "shuffle": False,
    "repeat": False,
    "max_length": 56,
    "special_tokens_already_in_vocab": False,
  },
}

# inference as in en-de-nmt-small.py: the same variables under the beam-search decoder
from open_seq2seq.decoders import BeamSearchRNNDecoderWithAttention

infer_params = {
  "batch_size_per_gpu": 8,
  "decoder": BeamSearchRNNDecoderWithAttention,
  "decoder_params": {
    "beam_width": 10,
    "length_penalty": 1.0,
    "core_cell": tf.nn.rnn_cell.LSTMCell,
    "core_cell_params": {"num_units": 512, "forget_bias": 1.0},
    "decoder_layers": 2,
    "decoder_dp_input_keep_prob": 0.8,
    "decoder_dp_output_keep_prob": 1.0,
    "decoder_use_skip_connections": False,
    "GO_SYMBOL": SpecialTextTokens.S_ID.value,
    "END_SYMBOL": SpecialTextTokens.EOS_ID.value,
    "PAD_SYMBOL": SpecialTextTokens.PAD_ID.value,
    "tgt_emb_size": 512,
    "attention_type": "gnmt_v2",
    "attention_layer_size": 512,
  },
  "data_layer": ParallelTextDataLayer,
  "data_layer_params": {
    "src_vocab_file": data_root + "vocab/source.txt",
    "tgt_vocab_file": data_root + "vocab/target.txt",
    "source_file": data_root + "test/source.txt",
    "target_file": data_root + "test/source.txt",   # unused by infer
    "delimiter": " ",
    "shuffle": False,
    "repeat": False,
    "max_length": 256,
    "special_tokens_already_in_vocab": False,
  },
}
