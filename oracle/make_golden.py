"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/*.npz from the real reference.

Run in the build container only (needs /root/reference, which never travels to the
GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

It imports the reference package read-only, runs every hot-path layer
with ``np.random.seed`` fixed, and stores inputs, initial params, outputs, input
grads and post-SGD params.  The fixtures are DATA (arrays), not source.  Each file
records ``numpy_version`` because the reference's dtypes depend on it (NEP 50).
"""

import io
import os
import sys
import contextlib

import numpy as np

REFERENCE = os.environ.get('NPM_REFERENCE', '/root/reference')
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')


def _ref():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REFERENCE)
    import layers, optimizer, loss, train  # noqa: E401
    return layers, optimizer, loss, train


def rand(shape):
    return np.random.normal(size=shape).astype(np.float32)


def save(name, **arrays):
    arrays['numpy_version'] = np.array(np.__version__)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrays)
    print(f'{name}: {os.path.getsize(path) / 1024:.0f} KiB')


def gen_dense(layers, with_relu, name):
    np.random.seed(0)
    layer = layers.Dense(units=16) if with_relu else layers.Linear(units=16)
    x = rand([64, 32])
    y = layer(x)
    lin = layer.linear if with_relu else layer
    w0, b0 = lin.w.copy(), lin.b.copy()
    dy = rand([64, 16])
    dx = layer(dy, backprop=True, learning_rate=0.01)
    save(name, x=x, w0=w0, b0=b0, y=y, dy=dy, dx=dx, w1=lin.w, b1=lin.b, lr=np.float64(0.01))


def gen_activations(layers):
    np.random.seed(0)
    x = rand([48, 40])
    dy = rand([48, 40])
    relu = layers.ReLU()
    x[0, :5] = 0.0                      # exercise the x == 0 branch of the >= test
    y = relu(x)
    dx = relu.backward(dy)
    save('relu', x=x, y=y, dy=dy, dx=dx)
    np.random.seed(1)
    sm = layers.Softmax()
    x = rand([6, 24, 40]) * 3.0
    dy = rand([6, 24, 40])
    y = sm(x)
    dx = sm(dy, backprop=True)
    save('softmax', x=x, y=y, dy=dy, dx=dx)


def gen_layernorm(layers, shape, eps, name):
    np.random.seed(0)
    ln = layers.LayerNormalization() if eps is None else layers.LayerNormalization(epsilon=eps)
    x = rand(shape) * 2.0 + 0.5
    z = ln(x)
    g0, b0 = ln._gamma.copy(), ln._beta.copy()
    dz = rand(shape)
    dx = ln(dz, backprop=True, learning_rate=1e-3)
    save(name, x=x, gamma0=g0, beta0=b0, eps=np.float64(ln._epsilon), z=z, dz=dz, dx=dx,
         gamma1=ln._gamma, beta1=ln._beta, lr=np.float64(1e-3))


def gen_conv(layers, shape, channels, k, name):
    np.random.seed(0)
    layer = layers.Conv2D(channels=channels, kernel_size=k)
    x = rand(shape)
    y = layer(x)
    w0, b0 = layer.w.copy(), layer.b.copy()
    dy = rand(y.shape)
    dx = layer(dy, backprop=True, learning_rate=0.01)
    save(name, x=x, w0=w0, b0=b0, y=y, dy=dy, dx=dx, w1=layer.w, b1=layer.b, lr=np.float64(0.01))


def gen_mha(layers, batch, sq, skv, feat, heads, name):
    np.random.seed(0)
    layer = layers.MultiHeadAttention(num_heads=heads)
    query = rand([batch, sq, feat])
    cross = skv is not None
    kv = rand([batch, skv, feat]) if cross else None
    out = layer(query, kv) if cross else layer(query)
    names = ['wq', 'wk', 'wv', 'wo', 'bq', 'bk', 'bv', 'bo']
    p0 = {f'{n}0': getattr(layer, '_' + n).copy() for n in names}
    dy = rand(out.shape) * 0.05
    dq, dk, dv = layer(dy, backprop=True, learning_rate=0.01)
    p1 = {f'{n}1': getattr(layer, '_' + n) for n in names}
    extra = dict(kv=kv) if cross else {}
    save(name, query=query, out=out, dy=dy, dquery=dq, dkey=dk, dvalue=dv,
         heads=np.int64(heads), lr=np.float64(0.01), **p0, **p1, **extra)


def gen_encoder(layers, norm_first, name, drop_rate=0.0):
    """drop_rate > 0: the masks the reference's DropOut layers drew from the global generator (in between the lazy parameter
    draws of the first call) are stored too; a seeded run of the build must draw the very same ones."""
    np.random.seed(0)
    enc = layers.TransformerEncoder(num_heads=4, hidden_units=96, norm_first=norm_first, drop_rate=drop_rate)
    qkv = rand([3, 16, 48])
    out = enc(qkv)
    att = enc._self_attention
    subs = {'att_' + n: (att, '_' + n) for n in ['wq', 'wk', 'wv', 'wo', 'bq', 'bk', 'bv', 'bo']}
    subs.update(n1_gamma=(enc._norm1, '_gamma'), n1_beta=(enc._norm1, '_beta'),
                n2_gamma=(enc._norm2, '_gamma'), n2_beta=(enc._norm2, '_beta'),
                d1_w=(enc._dense1._linear, '_w'), d1_b=(enc._dense1._linear, '_b'),
                d2_w=(enc._dense2, '_w'), d2_b=(enc._dense2, '_b'))
    p0 = {k + '__0': getattr(o, a).copy() for k, (o, a) in subs.items()}
    dy = rand(out.shape) * 0.05
    dx = enc(dy, backprop=True, learning_rate=1e-3)
    p1 = {k + '__1': getattr(o, a) for k, (o, a) in subs.items()}
    extra = {}
    if drop_rate:
        extra = dict(drop_rate=np.float64(drop_rate), mask1=np.asarray(enc._dropout1._mask).astype(np.uint8),
                     mask2=np.asarray(enc._dropout2._mask).astype(np.uint8))
    save(name, qkv=qkv, out=out, dy=dy, dx=dx, norm_first=np.bool_(norm_first),
         heads=np.int64(4), hidden=np.int64(96), lr=np.float64(1e-3), **p0, **p1, **extra)


def gen_decoder(layers, norm_first, name, drop_rate=0.0):
    np.random.seed(0)
    dec = layers.TransformerDecoder(num_heads=4, hidden_units=80, norm_first=norm_first, drop_rate=drop_rate)
    q = rand([3, 12, 48])
    kv = rand([3, 20, 48])
    out = dec(q, kv)
    names = ['wq', 'wk', 'wv', 'wo', 'bq', 'bk', 'bv', 'bo']
    subs = {'sa_' + n: (dec._self_attention, '_' + n) for n in names}
    subs.update({'ca_' + n: (dec._cross_attention, '_' + n) for n in names})
    for i, norm in enumerate([dec._norm1, dec._norm2, dec._norm3], start=1):
        subs[f'n{i}_gamma'] = (norm, '_gamma')
        subs[f'n{i}_beta'] = (norm, '_beta')
    subs.update(d1_w=(dec._dense1._linear, '_w'), d1_b=(dec._dense1._linear, '_b'),
                d2_w=(dec._dense2, '_w'), d2_b=(dec._dense2, '_b'))
    p0 = {k + '__0': getattr(o, a).copy() for k, (o, a) in subs.items()}
    dy = rand(out.shape) * 0.05
    dq, dkv = dec(dy, backprop=True, learning_rate=1e-3)
    p1 = {k + '__1': getattr(o, a) for k, (o, a) in subs.items()}
    extra = {}
    if drop_rate:
        extra = dict(drop_rate=np.float64(drop_rate), **{f'mask{i}': np.asarray(d._mask).astype(np.uint8)
                                                        for i, d in enumerate([dec._dropout1, dec._dropout2, dec._dropout3], start=1)})
    save(name, q=q, kv=kv, out=out, dy=dy, dq=dq, dkv=dkv, norm_first=np.bool_(norm_first), heads=np.int64(4),
         hidden=np.int64(80), lr=np.float64(1e-3), **p0, **p1, **extra)


def gen_losses(loss):
    """loss_test.py:15-66 shapes: [128, 32]."""
    np.random.seed(0)
    y = rand([128, 32])
    t = rand([128, 32])
    mse = loss.MSELoss()
    value = mse(y, t)
    grad = mse(backprop=True)
    prob = np.exp(y) / np.exp(y).sum(axis=-1, keepdims=True)
    onehot = np.eye(32, dtype=np.float32)[np.random.randint(0, 32, size=128)]
    ce = loss.CrossEntropyLoss()
    ce_value = ce(prob, onehot)
    ce_grad = ce(backprop=True)
    save('losses', y=y, t=t, mse=np.float64(value), mse_grad=grad, prob=prob, onehot=onehot,
         ce=np.float64(ce_value), ce_grad=ce_grad)


def gen_softmax_ce(layers, loss):
    """loss_test.py:49-66: the composed flow ``ce(softmax(y), t)`` forward and
    ``softmax(ce(y, t, backprop=True), backprop=True)`` backward, with soft targets (a softmax of noise, as the test
    draws them).  Note the call form: the loss's backward ignores its positional arguments and uses what its
    forward cached (loss.py:33-39)."""
    np.random.seed(0)
    y = rand([128, 32])
    noise = rand([128, 32])
    targets = (np.exp(noise) / np.exp(noise).sum(axis=-1, keepdims=True)).astype(np.float32)
    ce = loss.CrossEntropyLoss()
    softmax = layers.Softmax()
    prob = softmax(y)
    value = ce(prob, targets)
    dprob = ce(y, targets, backprop=True)
    dy = softmax(dprob, backprop=True)
    save('softmax_ce', y=y, targets=targets, prob=prob, ce=np.float64(value), dprob=dprob, dy=dy)


def gen_train(layers, optimizer, train):
    """train_test.py:14-49 flow; the printed losses are the known answers."""
    import re
    for opt_name in ('sgd', 'adam'):
        np.random.seed(0)
        feats = [16, 32, 64, 32, 16]
        stack = [layers.Dense(units=f, name=f'layer_{i}') for i, f in enumerate(feats)]
        x = np.random.uniform(0.0, 1.0, size=[128, 16]).astype(np.float32)
        t = np.random.uniform(0.0, 1.0, size=[128, 16]).astype(np.float32)
        opt = optimizer.AdamOptimizer(1e-4) if opt_name == 'adam' else optimizer.SGDOptimizer(1e-4)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            tr = train.Trainer(stack)
            tr.train(inputs=x, targets=t, steps=10, optimizer_=opt)
            tr.eval(inputs=x, targets=t)
        losses = np.array([float(v) for v in re.findall(r'Loss:\s+([0-9.eE+-]+)', buf.getvalue())])
        assert losses.size == 11
        final = {f'w{i}': l.linear.w for i, l in enumerate(stack)}
        save('train_mlp_' + opt_name, x=x, targets=t, losses=losses, lr=np.float64(1e-4), **final)


def gen_refshape(layers, name):
    """Fixtures at the reference's own test shapes and at the head sizes the fused attention kernels take
    (recipes, flow and rationale: tests/refshapes.py).  Stored: the reference's outputs as float32, the raw gradients
    its backward hands to ``optimizer_.update`` (a recording optimizer), the parameters after its own SGD step where
    the reference's tests assert them (MHA, Conv2D), and CRC-32s of the rebuilt inputs / parameters."""
    import copy
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
    import refshapes as R
    case = R.CASES[name]
    kind = case['kind']
    inp = R.draw_inputs(case)
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    arrays = {k + '_crc': np.int64(R.crc(v)) for k, v in inp.items()}
    arrays['lr'] = np.float64(case.get('lr', 0.0))
    if kind == 'conv':
        layer = layers.Conv2D(channels=case['channels'], kernel_size=case['k'])
        y = layer(inp['x'])
        arrays.update(w0=layer.w.copy(), b0=layer.b.copy(), y=f32(y))
        dy = R.mse_grad(y, inp['targets'])
        rec, recorder = copy.deepcopy(layer), R.GradRecorder()
        dx = rec(dy, backprop=True, optimizer_=recorder)
        arrays.update(dx_samples=f32(dx[list(R.CONV_DX_SAMPLES)]), dx_batch_sum=np.asarray(dx, dtype=np.float64).sum(axis=0),
                      dw=f32(recorder.grads[(id(rec), '_w')]), db=f32(recorder.grads[(id(rec), '_b')]))
        upd = copy.deepcopy(layer)
        upd(dy, backprop=True, learning_rate=case['lr'])
        arrays.update(w1=upd.w, b1=upd.b, dy_crc=np.int64(R.crc(dy)))
        save(name, **arrays)
        return
    if kind == 'dense':
        layer = layers.Dense(units=case['units'])
        y = layer(inp['x'])
        lin = layer.linear
        arrays.update(w0=lin.w.copy(), b0=lin.b.copy(), y=f32(y))
        dy = R.mse_grad(y, inp['targets'])
        rec, recorder = copy.deepcopy(layer), R.GradRecorder()
        dx = rec(dy, backprop=True, optimizer_=recorder)
        arrays.update(dx=f32(dx), dw=f32(recorder.grads[(id(rec.linear), '_w')]), db=f32(recorder.grads[(id(rec.linear), '_b')]))
        layer(dy, backprop=True, learning_rate=case['lr'])
        arrays.update(w1=lin.w, b1=lin.b, dy_crc=np.int64(R.crc(dy)))
        save(name, **arrays)
        return
    if kind == 'softmax':
        layer = layers.Softmax()
        y = layer(inp['x'])
        dy = R.mse_grad(y, inp['targets'])
        arrays.update(y=f32(y), dx=f32(layer(dy, backprop=True)), dy_crc=np.int64(R.crc(dy)))
        save(name, **arrays)
        return
    if kind == 'layernorm':
        layer = layers.LayerNormalization()
        layer(inp['x'])
        params = R.bound_params(case)
        layer._gamma, layer._beta, layer._epsilon = params['gamma'].copy(), params['beta'].copy(), R.LN_EPS
        z = layer(inp['x'])
        dz = R.mse_grad(z, inp['targets'])
        rec, recorder = copy.deepcopy(layer), R.GradRecorder()
        dx = rec(dz, backprop=True, optimizer_=recorder)
        arrays.update(z=f32(z), dx=f32(dx), dgamma=f32(recorder.grads[(id(rec), '_gamma')]), dbeta=f32(recorder.grads[(id(rec), '_beta')]))
        layer(dz, backprop=True, learning_rate=case['lr'])
        arrays.update(gamma1=f32(layer._gamma), beta1=f32(layer._beta), dy_crc=np.int64(R.crc(dz)))
        save(name, **arrays)
        return
    if kind == 'mha':
        layer = layers.MultiHeadAttention(num_heads=case['heads'])
    elif kind == 'encoder':
        layer = layers.TransformerEncoder(num_heads=case['heads'], hidden_units=case['hidden'], norm_first=case['norm_first'])
    else:
        layer = layers.TransformerDecoder(num_heads=case['heads'], hidden_units=case['hidden'], norm_first=case['norm_first'])
    args = (inp['query'], inp['kv']) if 'kv' in inp else (inp['query'],)
    layer(*args)                                   # lazy initialisation; these draws are replaced below
    params = R.bound_params(case)
    R.bind(layer, case, params)
    out = layer(*args)
    dy = R.mse_grad(out, inp['targets'])
    arrays.update(out=f32(out), dy_crc=np.int64(R.crc(dy)), params_crc=np.int64(R.crc(np.concatenate([v.ravel() for v in params.values()]))))
    rec, recorder = copy.deepcopy(layer), R.GradRecorder()
    grads_in = rec(dy, backprop=True, optimizer_=recorder)
    if kind == 'mha':
        arrays.update(dquery=f32(grads_in[0]), dkey=f32(grads_in[1]), dvalue=f32(grads_in[2]))
    elif kind == 'encoder':
        arrays.update(dx=f32(grads_in))
    else:
        arrays.update(dq=f32(grads_in[0]), dkv=f32(grads_in[1]))
    if kind == 'mha':          # (the reference's encoder / decoder tests assert outputs and input gradients only; the parameter
        #                        gradients of the composites are pinned at small shapes by encoder_*.npz / decoder_*.npz)
        arrays.update({'grad_' + k: f32(v) for k, v in recorder.named(rec, case).items()})
    if case.get('updated'):                        # attentions_test.py:72-85 asserts the updated parameters
        upd = copy.deepcopy(layer)
        upd(dy, backprop=True, learning_rate=case['lr'])
        arrays.update({n + '1': f32(getattr(upd, '_' + n)) for n in R.MHA_NAMES})
    save(name, **arrays)


def main():
    os.makedirs(OUT, exist_ok=True)
    layers, optimizer, loss, train = _ref()
    gen_dense(layers, True, 'dense')
    gen_dense(layers, False, 'linear')
    gen_activations(layers)
    gen_layernorm(layers, [32, 128], 1e-6, 'layernorm_2d')
    gen_layernorm(layers, [3, 10, 72], None, 'layernorm_3d')
    gen_conv(layers, [3, 9, 7, 8], 12, 3, 'conv_k3')
    gen_conv(layers, [2, 8, 8, 4], 6, 5, 'conv_k5')
    gen_conv(layers, [2, 6, 5, 8], 4, 1, 'conv_k1')
    gen_mha(layers, 4, 24, None, 64, 8, 'mha_self')
    gen_mha(layers, 3, 10, 28, 48, 4, 'mha_cross')
    gen_encoder(layers, True, 'encoder_prenorm')
    gen_encoder(layers, False, 'encoder_postnorm')
    gen_encoder(layers, True, 'encoder_dropout_prenorm', drop_rate=0.1)
    gen_encoder(layers, False, 'encoder_dropout_postnorm', drop_rate=0.1)
    gen_decoder(layers, True, 'decoder_prenorm')
    gen_decoder(layers, False, 'decoder_postnorm')
    gen_decoder(layers, True, 'decoder_dropout_prenorm', drop_rate=0.1)
    gen_decoder(layers, False, 'decoder_dropout_postnorm', drop_rate=0.1)
    gen_losses(loss)
    gen_softmax_ce(layers, loss)
    gen_train(layers, optimizer, train)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
    import refshapes
    for name in refshapes.CASES:
        gen_refshape(layers, name)


if __name__ == '__main__':
    main()
