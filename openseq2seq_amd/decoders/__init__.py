from .decoder import Decoder
from .fc_decoders import FullyConnectedTimeDecoder, FullyConnectedCTCDecoder
from .transformer_decoder import TransformerDecoder
from .rnn_decoders import RNNDecoderWithAttention
from .tacotron2_decoder import Tacotron2Decoder
