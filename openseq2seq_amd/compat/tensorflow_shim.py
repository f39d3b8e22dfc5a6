"""A `tensorflow`-named token module for loading the reference's config files.

The configs do `import tensorflow as tf` only to NAME things — dtypes, activation
functions, initializers, regularizers, optimizers (`tf.nn.relu`,
`tf.contrib.layers.xavier_initializer`, `tf.float16`, `tf.contrib.opt.LazyAdamOptimizer`,
...). This module provides exactly those names as inert tokens whose
`__name__` the host layer dispatches on. It performs no computation and is installed
under `sys.modules['tensorflow']` only while a config is being loaded and only when the
real TensorFlow is not importable (utils/utils.py: load_config_module)."""
import types


class _Token(object):
  def __init__(self, name):
    self.__name__ = name
    self.name = name

  def __call__(self, *a, **kw):   # e.g. l2_regularizer(scale) -> parametrised token
    t = _Token(self.__name__)
    t.args, t.kwargs = a, kw
    return t

  def __repr__(self):
    return "<tf token %s>" % self.__name__


def _ns(name, **members):
  m = types.ModuleType(name)
  for k, v in members.items():
    setattr(m, k, v)
  return m


float16 = _Token("float16")
float32 = _Token("float32")
float64 = _Token("float64")
int32 = _Token("int32")
int64 = _Token("int64")
bfloat16 = _Token("bfloat16")

minimum = _Token("minimum")
maximum = _Token("maximum")
glorot_uniform_initializer = _Token("glorot_uniform_initializer")
random_normal_initializer = _Token("random_normal_initializer")
random_uniform_initializer = _Token("random_uniform_initializer")
truncated_normal_initializer = _Token("truncated_normal_initializer")
constant_initializer = _Token("constant_initializer")

nn = _ns("tensorflow.nn", relu=_Token("relu"), tanh=_Token("tanh"), sigmoid=_Token("sigmoid"),
         relu6=_Token("relu6"), elu=_Token("elu"), softmax=_Token("softmax"),
         rnn_cell=_ns("tensorflow.nn.rnn_cell", LSTMCell=_Token("LSTMCell"),
                      GRUCell=_Token("GRUCell"), BasicLSTMCell=_Token("BasicLSTMCell")))
train = _ns("tensorflow.train", AdamOptimizer=_Token("AdamOptimizer"),
            MomentumOptimizer=_Token("MomentumOptimizer"),
            GradientDescentOptimizer=_Token("GradientDescentOptimizer"),
            RMSPropOptimizer=_Token("RMSPropOptimizer"))
layers = _ns("tensorflow.layers")
contrib = _ns(
    "tensorflow.contrib",
    layers=_ns("tensorflow.contrib.layers", xavier_initializer=_Token("xavier_initializer"),
               l2_regularizer=_Token("l2_regularizer"),
               variance_scaling_initializer=_Token("variance_scaling_initializer")),
    opt=_ns("tensorflow.contrib.opt", LazyAdamOptimizer=_Token("LazyAdamOptimizer"),
            AdamWOptimizer=_Token("AdamWOptimizer")),
    cudnn_rnn=_ns("tensorflow.contrib.cudnn_rnn", CudnnLSTM=_Token("CudnnLSTM"),
                  CudnnGRU=_Token("CudnnGRU")),
    rnn=_ns("tensorflow.contrib.rnn", LSTMCell=_Token("LSTMCell")),
)
keras = _ns("tensorflow.keras", initializers=_ns("tensorflow.keras.initializers",
                                                 Ones=_Token("Ones"), Zeros=_Token("Zeros")))
__version__ = "1.13.1-os2s-shim"
