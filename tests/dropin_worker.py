"""TEST INFRASTRUCTURE (build container only): the reference's OWN, unchanged ``optimizer.py``, ``loss.py`` and
``train.py`` driving the device layers after ``np_modeling_amd.install()`` -- the drop-in claim of
BASELINE.json's north_star ("Trainer/optimizer/loss drop in unchanged"), on the host simulator of the C ABI.

Run as a child process by tests/test_reference_dropin.py (a fresh interpreter: the reference's module names
``optimizer`` / ``loss`` / ``train`` must not already be bound to anything else).  Flows:

* reference train_test.py:14-49 -- MLP 16 -> [16, 32, 64, 32, 16], batch 128, 10 steps, SGD(1e-4) and Adam(1e-4),
  printed losses and final weights against tests/golden/train_mlp_*.npz (made from the reference itself);
* reference train_test.py:51-81 -- Conv2D stack k = [1, 3, 5, 3, 1], SGD(1e-6), 2 steps, against the oracle;
* reference optimizer.py:26-33,36-69 stepping a TransformerEncoder backward (SGD against the golden post-step
  parameters; Adam against the oracle's restatement of the update rule on the gradients the layers produce).

Prints ``DROPIN OK`` on success."""

import contextlib
import io
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.dont_write_bytecode = True

REFERENCE = os.environ.get('NPM_REFERENCE', '/root/reference')


def main():
    import hostsim
    import np_modeling_amd as npm
    from conftest import assert_close, load_golden
    from oracle import np_oracle as O

    hostsim.install()
    npm.install()                               # `layers` now resolves to the device layers
    sys.path.insert(0, REFERENCE)
    import loss                                 # the reference's own modules
    import optimizer
    import train
    from layers import conv, mlp, transformer   # ours, through the reference's import names
    for mod in (loss, optimizer, train):
        assert os.path.realpath(mod.__file__).startswith(os.path.realpath(REFERENCE)), mod.__file__
    assert mlp.Dense is npm.layers.Dense and conv.Conv2D is npm.layers.Conv2D

    def losses_of(text):
        return np.array([float(v) for v in re.findall(r'Loss:\s+([0-9.eE+-]+)', text)])

    # ---- train_test.py:14-49 -------------------------------------------------------------------------------
    for name, make in (('sgd', lambda: optimizer.SGDOptimizer(1e-4)), ('adam', lambda: optimizer.AdamOptimizer(1e-4))):
        g = load_golden('train_mlp_' + name)
        np.random.seed(0)
        stack = [mlp.Dense(units=f, name=f'layer_{i}') for i, f in enumerate([16, 32, 64, 32, 16])]
        x = np.random.uniform(0.0, 1.0, size=[128, 16]).astype(np.float32)
        t = np.random.uniform(0.0, 1.0, size=[128, 16]).astype(np.float32)
        np.testing.assert_array_equal(x, g['x'])
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            trainer = train.Trainer(stack)
            trainer.train(inputs=x, targets=t, steps=10, optimizer_=make())
            trainer.eval(inputs=x, targets=t)
        got = losses_of(buf.getvalue())
        np.testing.assert_allclose(got, g['losses'], rtol=2e-6, err_msg=name)
        for i, layer in enumerate(stack):
            assert_close(layer.linear.w, g[f'w{i}'], tol=2e-6, what=f'{name} w{i}')
        print(f'train_mlp[{name}]: max rel loss error {np.abs(got / g["losses"] - 1).max():.1e}')

    # ---- train_test.py:51-81 (2 steps) ---------------------------------------------------------------------
    np.random.seed(0)
    ks, cs = [1, 3, 5, 3, 1], [16, 32, 64, 32, 16]
    stack = [conv.Conv2D(channels=c, kernel_size=k, name=f'layer_{i}') for i, (c, k) in enumerate(zip(cs, ks))]
    x = np.random.uniform(-1.0, 1.0, size=[16, 32, 32, 16]).astype(np.float32)
    t = np.random.uniform(0.0, 1.0, size=[16, 32, 32, 16]).astype(np.float32)
    state = np.random.get_state()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        train.Trainer(stack).train(inputs=x, targets=t, steps=2, optimizer_=optimizer.SGDOptimizer(1e-6))
    got = losses_of(buf.getvalue())
    np.random.set_state(state)
    params, cin = [], 16
    for c, k in zip(cs, ks):
        params.append([O.random_init([k, k, cin, c]).astype(np.float64), O.random_init([c]).astype(np.float64)])
        cin = c
    want = []
    for _ in range(2):
        acts, pres = [x.astype(np.float64)], []
        for wgt, b in params:
            y, pre = O.conv_layer_fwd(acts[-1], wgt, b)
            acts.append(y)
            pres.append(pre)
        want.append(O.mse_fwd(acts[-1], t))
        dy = O.mse_bwd(acts[-1], t)
        for i in reversed(range(5)):
            dy, dw, db = O.conv_layer_bwd(acts[i], params[i][0], pres[i], dy)
            params[i][0] = params[i][0] - 1e-6 * dw
            params[i][1] = params[i][1] - 1e-6 * db
    np.testing.assert_allclose(got, want, rtol=1e-5)
    for i, layer in enumerate(stack):
        assert_close(layer.w, params[i][0], tol=1e-5, what=f'conv w{i}')
    print('train_conv: losses', got)

    # ---- reference optimizers stepping an encoder backward ---------------------------------------------------
    g = load_golden('encoder_prenorm')
    lr = float(g['lr'])
    keys = [k[:-3] for k in g if k.endswith('__0')]

    def fresh():
        np.random.seed(0)
        enc = transformer.TransformerEncoder(num_heads=int(g['heads']), hidden_units=int(g['hidden']), norm_first=True)
        out = enc(np.random.normal(size=g['qkv'].shape).astype(np.float32))      # == g['qkv'] (seeded)
        assert_close(out, g['out'], tol=1e-5)
        return enc

    def param(enc, key):
        group, attr = key.split('_', 1)
        owner = {'att': enc._self_attention, 'n1': enc._norm1, 'n2': enc._norm2,
                 'd1': enc._dense1.linear, 'd2': enc._dense2}[group]
        return np.asarray(getattr(owner, '_' + attr))

    enc = fresh()
    ref_sgd = optimizer.SGDOptimizer(lr)         # the REFERENCE's class, unchanged (optimizer.py:26-33)
    seen = []
    inner = ref_sgd.update_variable
    ref_sgd.update_variable = lambda identifier, variable, gradient: (seen.append(identifier), inner(identifier, variable, gradient))[1]
    import np_modeling_amd
    sim = np_modeling_amd._C._LIB
    before_axpy = sim.calls.count('npm_axpy')
    dx = enc(g['dy'], backprop=True, optimizer_=ref_sgd)
    assert_close(dx, g['dx'], tol=1e-5)
    for key in keys:
        assert_close(param(enc, key), g[key + '__1'], tol=1e-5, what='sgd ' + key)
    # Optimizer.update(obj, attribute, grad) ran once per parameter with the reference's id(obj).attribute keys -- and the 16
    # ``variable -= lr * gradient`` of optimizer.py:32 were ONE device launch (parameters in one arena, gradients in the
    # bucket that mirrors it: SURVEY.md section 8f rank 1)
    assert len(seen) == 16 and len(set(seen)) == 16 and all('.' in s and s.split('.')[0].isdigit() for s in seen), seen
    assert sim.calls.count('npm_axpy') - before_axpy == 1, sim.calls.count('npm_axpy') - before_axpy

    class Recorder:                              # the gradients the layers hand to an optimizer, parameters untouched
        def __init__(self):
            self.grads = {}

        def update(self, obj, attribute, gradient):
            self.grads[(id(obj), attribute)] = np.asarray(gradient).astype(np.float64)

    enc, rec = fresh(), Recorder()
    enc(g['dy'], backprop=True, optimizer_=rec)
    owners = {'att': enc._self_attention, 'n1': enc._norm1, 'n2': enc._norm2, 'd1': enc._dense1.linear, 'd2': enc._dense2}
    before = {key: param(enc, key).astype(np.float64) for key in keys}
    enc(g['qkv'])                                # the same forward again (the recorder changed nothing)
    enc(g['dy'], backprop=True, optimizer_=optimizer.AdamOptimizer(lr))
    for key in keys:
        group, attr = key.split('_', 1)
        grad = rec.grads[(id(owners[group]), '_' + attr)]
        want_p = O.adam_step(before[key], grad, {}, lr)
        # Adam's first step moves an element by about lr (less where |g| is below sqrt(eps)); fp32 parameters
        # of magnitude <= 1 round to 6e-8
        np.testing.assert_allclose(param(enc, key), want_p, rtol=0, atol=2e-4 * lr + 1.2e-7, err_msg='adam ' + key)
    print('encoder: reference SGD and Adam steps match')
    print('DROPIN OK')


if __name__ == '__main__':
    main()
