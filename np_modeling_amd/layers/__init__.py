"""Device layers under the names of the reference's ``layers`` package (reference layers/__init__.py:1-7), plus
``DropOut``, which the reference keeps in ``layers.normalizations`` only."""

from np_modeling_amd.layers import activations, attentions, conv, layer, mlp, normalizations, transformer

Layer = layer.Layer
Activation, ReLU, Softmax = activations.Activation, activations.ReLU, activations.Softmax
Linear, Dense = mlp.Linear, mlp.Dense
Conv2D = conv.Conv2D
LayerNormalization, DropOut = normalizations.LayerNormalization, normalizations.DropOut
MultiHeadAttention = attentions.MultiHeadAttention
TransformerEncoder, TransformerDecoder = transformer.TransformerEncoder, transformer.TransformerDecoder

__all__ = ['Layer', 'Activation', 'ReLU', 'Softmax', 'Linear', 'Dense', 'Conv2D', 'LayerNormalization', 'DropOut',
           'MultiHeadAttention', 'TransformerEncoder', 'TransformerDecoder']
