"""Checkpoint interchange with the reference.

  read_checkpoint(path)        this build's pickles AND stable-baselines 2.8 pickles such as the reference's
                               IRRL/script/pkl/bp5_155.pkl: `(data dict, [19 np arrays])` written by
                               PPO2.save -> _save_to_file(cloudpickle=True) (ppo2.py:452-476).  The SB pickle
                               embeds tensorflow / gym / cloudpickle objects; a stub unpickler replaces them so
                               neither package is needed (only the hyper-parameters and the arrays are kept).
  export_actor_csv(model, dir) the reference's `--o` export (CustomerLstmNN.save_model, NN:203-224):
                               lstm_wx{i}.csv, lstm_wh{i}.csv, lstm_b{i}.csv, pi_w.csv, pi_b.csv with '%.6f'.
  NumpyLstmActor               numpy twin of CustomerLstmNN.predict (NN:112-135) for deployment checks.
"""
import os
import pickle

import numpy as np


class _Stub(object):
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Stub()

    def __setstate__(self, s):
        self.state = s

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Stub()


def _stub_fn(*a, **k):
    return _Stub()


class _StubUnpickler(pickle.Unpickler):
    """ALLOW-LIST unpickler: a checkpoint only ever needs numpy arrays, plain containers and scalars.  Exactly the globals
    below resolve to the real thing; every other global a pickle names -- the tensorflow / gym / cloudpickle / stable_baselines
    objects of a reference checkpoint, but also `os.system`, `builtins.eval` or anything else a crafted file could ask for --
    becomes an inert stub that ignores its arguments.  Loading a checkpoint can therefore not run foreign code."""
    _ALLOWED = {
        ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
        ("numpy._core.multiarray", "scalar"), ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.numeric", "_frombuffer"),
        ("numpy._core.numeric", "_frombuffer"), ("numpy", "float32"), ("numpy", "float64"), ("numpy", "int32"), ("numpy", "int64"), ("numpy", "bool_"),
        ("collections", "OrderedDict"), ("_codecs", "encode"),
        ("builtins", "list"), ("builtins", "dict"), ("builtins", "tuple"), ("builtins", "set"), ("builtins", "frozenset"), ("builtins", "int"),
        ("builtins", "float"), ("builtins", "bool"), ("builtins", "str"), ("builtins", "bytes"), ("builtins", "bytearray"), ("builtins", "complex"),
        ("builtins", "slice"), ("builtins", "range"), ("__builtin__", "list"), ("__builtin__", "dict"), ("__builtin__", "tuple"), ("__builtin__", "set"),
    }
    _FNS = ("CodeType", "code", "_make_skel_func", "_fill_function", "_builtin_type", "_make_cell", "_make_empty_cell",
            "_rehydrate_skeleton_class", "_make_skeleton_class", "subimport")

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        return _stub_fn if name in self._FNS else _Stub


def read_checkpoint(path):
    if not os.path.exists(path) and os.path.exists(path + ".pkl"):
        path = path + ".pkl"
    with open(path, "rb") as f:
        data, params = _StubUnpickler(f).load()
    if isinstance(params, dict):
        params = list(params.values())
    clean = {}
    for k, v in dict(data).items():
        clean[k] = v if isinstance(v, (int, float, str, list, dict, tuple, type(None))) else None
    return clean, [np.asarray(p) for p in params]


def export_actor_csv(model, out_dir):
    """NN:203-224.  Only the actor is exported, as the reference does."""
    os.makedirs(out_dir, exist_ok=True)
    params = model.get_parameter_list()
    n_layers = len(model.policy.n_lstm)
    for i in range(n_layers):
        np.savetxt(os.path.join(out_dir, "lstm_wx%d.csv" % i), params[3 * i + 0], fmt="%.6f", delimiter=",")
        np.savetxt(os.path.join(out_dir, "lstm_wh%d.csv" % i), params[3 * i + 1], fmt="%.6f", delimiter=",")
        np.savetxt(os.path.join(out_dir, "lstm_b%d.csv" % i), params[3 * i + 2], fmt="%.6f", delimiter=",")
    base = 6 * n_layers
    np.savetxt(os.path.join(out_dir, "pi_w.csv"), params[base + 2], fmt="%.6f", delimiter=",")
    np.savetxt(os.path.join(out_dir, "pi_b.csv"), params[base + 3], fmt="%.6f", delimiter=",")
    return out_dir


class NumpyLstmActor(object):
    """Stateful numpy forward of the LSTM actor (gate order i,f,o,g; output clipped to [-1,1], NN:133-134)."""

    def __init__(self, wx, wh, b, pi_w, pi_b):
        self.wx, self.wh, self.b, self.pi_w, self.pi_b = wx, wh, b, pi_w, pi_b
        self.reset()

    @classmethod
    def from_csv_dir(cls, d, n_layers=2):
        ld = lambda n: np.loadtxt(os.path.join(d, n), delimiter=",")
        return cls([ld("lstm_wx%d.csv" % i) for i in range(n_layers)], [ld("lstm_wh%d.csv" % i) for i in range(n_layers)],
                   [ld("lstm_b%d.csv" % i) for i in range(n_layers)], ld("pi_w.csv"), ld("pi_b.csv"))

    @classmethod
    def from_parameter_list(cls, params, n_layers=2):
        base = 6 * n_layers
        return cls([params[3 * i] for i in range(n_layers)], [params[3 * i + 1] for i in range(n_layers)],
                   [params[3 * i + 2] for i in range(n_layers)], params[base + 2], params[base + 3])

    def reset(self):
        self.c = [np.zeros(w.shape[0]) for w in self.wh]
        self.h = [np.zeros(w.shape[0]) for w in self.wh]

    def predict(self, obs):
        x = np.asarray(obs, np.float64)
        sig = lambda v: 1.0 / (1.0 + np.exp(-v))
        for i in range(len(self.wx)):
            n = self.wh[i].shape[0]
            z = x @ self.wx[i] + self.h[i] @ self.wh[i] + self.b[i]
            ig, fg, og, g = sig(z[:n]), sig(z[n:2 * n]), sig(z[2 * n:3 * n]), np.tanh(z[3 * n:])
            self.c[i] = fg * self.c[i] + ig * g
            self.h[i] = og * np.tanh(self.c[i])
            x = self.h[i]
        return np.clip(x @ self.pi_w + self.pi_b, -1.0, 1.0)
