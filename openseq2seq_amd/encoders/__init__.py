from .encoder import Encoder
from .tdnn_encoder import TDNNEncoder
from .transformer_encoder import TransformerEncoder
from .ds2_encoder import DeepSpeech2Encoder
from .rnn_encoders import (BidirectionalRNNEncoderWithEmbedding, UnidirectionalRNNEncoderWithEmbedding,
                           GNMTLikeEncoderWithEmbedding)
from .tacotron2_encoder import Tacotron2Encoder
