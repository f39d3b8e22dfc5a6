from .model import Model
from .encoder_decoder import EncoderDecoderModel
from .speech2text import Speech2Text
from .text2text import Text2Text
from .text2speech import Text2Speech, Text2SpeechTacotron
