# pylint: skip-file
"""The reference's tiny Transformer on the toy reversal task
(example_configs/text2text/toy-reversal/nmt-reversal-TT.py) with d_model raised from 128 to 512
(8 heads x 64): the HIP LayerNorm / attention kernels are built for the Transformer-base / -big
widths (hidden 512 / 1024, head dim 64). Everything else — 2+2 layers, LazyAdam +
transformer_policy with 200 warm-up steps, beam 5 / alpha 1.0 / extra_decode_length 2, label
smoothing default — is the reference's TT config, except learning_rate 1.0 -> 0.3: the policy's
peak rate scales as d_model^-0.5 only, and 1.0 (tuned for d_model 128) leaves the 4x wider
model at the unigram plateau (loss 2.2); 0.3 reaches BLEU 0.99 in 800 steps. Vocabularies are
padded to a multiple of 8 as in transformer-big.py."""
from __future__ import absolute_import, division, print_function
from open_seq2seq.models import Text2Text
from open_seq2seq.encoders import TransformerEncoder
from open_seq2seq.decoders import TransformerDecoder
from open_seq2seq.data.text2text.text2text import ParallelTextDataLayer
from open_seq2seq.losses import PaddedCrossEntropyLossWithSmoothing
from open_seq2seq.data.text2text.text2text import SpecialTextTokens
from open_seq2seq.optimizers.lr_policies import transformer_policy
import tensorflow as tf

base_model = Text2Text
d_model = 512
num_layers = 2
data_root = "toy_text_data/"

base_params = {
  "use_horovod": False,
  "num_gpus": 1,
  "batch_size_per_gpu": 64,
  "max_steps": 800,
  "print_loss_steps": 50,
  "eval_steps": 400,
  "logdir": "ReversalTask-Transformer-Transformer",
  "dtype": "mixed",
  "loss_scaling": "Backoff",

  "optimizer": tf.contrib.opt.LazyAdamOptimizer,
  "optimizer_params": {"beta1": 0.9, "beta2": 0.997, "epsilon": 0.000000001},
  "lr_policy": transformer_policy,
  "lr_policy_params": {"learning_rate": 0.3, "warmup_steps": 200, "d_model": d_model},

  "encoder": TransformerEncoder,
  "encoder_params": {
    "encoder_layers": num_layers, "hidden_size": d_model, "num_heads": 8,
    "attention_dropout": 0.1, "filter_size": 4 * d_model, "relu_dropout": 0.1,
    "layer_postprocess_dropout": 0.1, "remove_padding": True, "pad_embeddings_2_eight": True,
  },

  "decoder": TransformerDecoder,
  "decoder_params": {
    "layer_postprocess_dropout": 0.1, "num_hidden_layers": num_layers, "hidden_size": d_model,
    "num_heads": 8, "attention_dropout": 0.1, "relu_dropout": 0.1, "filter_size": 4 * d_model,
    "beam_size": 5, "alpha": 1.0, "extra_decode_length": 2,
    "EOS_ID": SpecialTextTokens.EOS_ID.value,
    "GO_SYMBOL": SpecialTextTokens.S_ID.value,
    "END_SYMBOL": SpecialTextTokens.EOS_ID.value,
    "PAD_SYMBOL": SpecialTextTokens.PAD_ID.value,
  },

  "loss": PaddedCrossEntropyLossWithSmoothing,
  "loss_params": {},
}

train_params = {
  "data_layer": ParallelTextDataLayer,
  "data_layer_params": {
    "pad_vocab_to_eight": True,
    "src_vocab_file": data_root + "vocab/source.txt",
    "tgt_vocab_file": data_root + "vocab/target.txt",
    "source_file": data_root + "train/source.txt",
    "target_file": data_root + "train/target.txt",
    "shuffle": True, "repeat": True, "max_length": 56, "delimiter": " ",
    "special_tokens_already_in_vocab": False,
  },
}

eval_params = {
  "data_layer": ParallelTextDataLayer,
  "data_layer_params": {
    "pad_vocab_to_eight": True,
    "src_vocab_file": data_root + "vocab/source.txt",
    "tgt_vocab_file": data_root + "vocab/target.txt",
    "source_file": data_root + "dev/source.txt",
    "target_file": data_root + "dev/target.txt",
    "shuffle": False, "repeat": False, "max_length": 56, "delimiter": " ",
    "special_tokens_already_in_vocab": False,
  },
}

infer_params = {
  "batch_size_per_gpu": 1,
  "data_layer": ParallelTextDataLayer,
  "data_layer_params": {
    "pad_vocab_to_eight": True,
    "src_vocab_file": data_root + "vocab/source.txt",
    "tgt_vocab_file": data_root + "vocab/source.txt",
    "source_file": data_root + "test/source.txt",
    "target_file": data_root + "test/target.txt",
    "shuffle": False, "repeat": False, "max_length": 256, "delimiter": " ",
    "special_tokens_already_in_vocab": False,
  },
}
