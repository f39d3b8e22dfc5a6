from .loss import Loss
from .ctc_loss import CTCLoss
from .sequence_loss import PaddedCrossEntropyLossWithSmoothing
from .sequence_loss import BasicSequenceLoss
from .text2speech_loss import Text2SpeechLoss
