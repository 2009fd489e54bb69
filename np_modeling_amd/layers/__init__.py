"""Same public surface as the reference's ``layers`` package (reference layers/__init__.py:1-7)."""

from np_modeling_amd.layers.activations import Activation, ReLU, Softmax
from np_modeling_amd.layers.attentions import MultiHeadAttention
from np_modeling_amd.layers.conv import Conv2D
from np_modeling_amd.layers.layer import Layer
from np_modeling_amd.layers.mlp import Dense, Linear
from np_modeling_amd.layers.normalizations import DropOut, LayerNormalization
from np_modeling_amd.layers.transformer import TransformerDecoder, TransformerEncoder
