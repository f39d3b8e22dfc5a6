"""Subword-tokenizer constants the configs import
(open_seq2seq/data/text2text/tokenizer.py: PAD_ID = 0, EOS_ID = 1)."""
PAD = "<pad>"
PAD_ID = 0
EOS = "<EOS>"
EOS_ID = 1
RESERVED_TOKENS = [PAD, EOS]
