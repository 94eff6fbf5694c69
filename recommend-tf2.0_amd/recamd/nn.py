"""Minimal Keras-shaped building blocks on top of recamd.ops.

The reference's plugin surface is `tf.keras.layers.Layer.__call__(inputs) -> call(inputs)` and
`tf.keras.Model.call` / `build_graph()`; TensorFlow is not available on either box, so the mirrored
classes in `ctr/` and `match/` derive from these instead.  They hold their weights EXPLICITLY (torch
tensors on the GPU) — the reference creates sub-layers inside `call()` during functional-graph
tracing (e.g. src/ctr/layers/modules.py:131, :255-269), so a native replacement has to own them.

Forward (inference) only: Dropout is the identity, BatchNormalization uses its moving statistics.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch

from . import ops


def default_device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("recamd needs a GPU: the HIP path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


_seed = [2020]  # mirrors random.seed(2020) of src/ctr/utils/data_process.py:11 (init values are
                # NOT parity-relevant: tests always load explicit weights)


def _gen(device):
    g = torch.Generator(device=device)
    g.manual_seed(_seed[0])
    _seed[0] += 1
    return g


def init_tensor(shape, initializer: str, device) -> torch.Tensor:
    """Keras initializer strings used by the reference."""
    t = torch.empty(shape, dtype=torch.float32, device=device)
    if initializer == "random_uniform":      # RandomUniform(-0.05, 0.05)
        t.uniform_(-0.05, 0.05, generator=_gen(device))
    elif initializer == "random_normal":     # RandomNormal(stddev=0.05)
        t.normal_(0.0, 0.05, generator=_gen(device))
    elif initializer == "glorot_uniform":    # Dense default kernel initializer
        fan_in, fan_out = shape[0], shape[-1]
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        t.uniform_(-lim, lim, generator=_gen(device))
    elif initializer == "zeros":
        t.zero_()
    elif initializer == "ones":
        t.fill_(1.0)
    else:
        raise ValueError(f"unknown initializer {initializer!r}")
    return t


class Layer:
    """Keras-Layer-shaped base: `layer(inputs)` -> `layer.call(inputs)`; weights in `self._w`."""

    def __init__(self, name: Optional[str] = None, device=None):
        self.name = name or type(self).__name__
        self._w: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self._children: "OrderedDict[str, Layer]" = OrderedDict()
        self._device = device
        self._v = 0  # bumped by set_weights (invalidates folded/cached weights)

    @property
    def _version(self) -> int:
        """cache key of everything derived from this layer's weights: set_weights bumps the layer's own counter,
        the optimiser kernels (raw-pointer writers) bump ops.weight_generation()"""
        from . import ops
        return self._v + ops.weight_generation()

    @_version.setter
    def _version(self, value: int) -> None:
        from . import ops
        self._v = value - ops.weight_generation()

    @property
    def device(self):
        if self._device is None:
            self._device = default_device()
        return self._device

    def add_weight(self, name: str, shape, initializer: str = "glorot_uniform") -> torch.Tensor:
        t = init_tensor(tuple(shape), initializer, self.device)
        self._w[name] = t
        return t

    def track(self, name: str, layer: "Layer") -> "Layer":
        self._children[name] = layer
        return layer

    def __setattr__(self, key, value):
        object.__setattr__(self, key, value)
        if isinstance(value, Layer) and key not in ("_children",) and "_children" in self.__dict__:
            self._children.setdefault(key, value)

    def __call__(self, inputs, *args, **kwargs):
        return self.call(inputs, *args, **kwargs)

    def call(self, inputs, **kwargs):  # pragma: no cover
        raise NotImplementedError

    # ---- explicit weight I/O (numpy), used by the parity tests ---------------------------------
    def get_weights(self) -> Dict[str, np.ndarray]:
        out = {k: v.detach().cpu().numpy() for k, v in self._w.items()}
        for cname, c in self._children.items():
            for k, v in c.get_weights().items():
                out[f"{cname}/{k}"] = v
        return out

    def set_weights(self, weights: Dict[str, np.ndarray]) -> None:
        for k, v in weights.items():
            head, _, rest = k.partition("/")
            if rest and head in self._children:
                self._children[head].set_weights({rest: v})
                continue
            if k not in self._w:
                raise KeyError(f"{self.name}: no weight named {k!r} (have {list(self._w)})")
            t = torch.as_tensor(np.asarray(v, np.float32)).to(self.device)
            if tuple(t.shape) != tuple(self._w[k].shape):
                raise ValueError(f"{self.name}/{k}: shape {tuple(t.shape)} != {tuple(self._w[k].shape)}")
            self._w[k].copy_(t)
        self._version += 1
        for c in self._children.values():
            c._version += 1

    def count_params(self) -> int:
        return sum(v.numel() for v in self._w.values()) + sum(c.count_params() for c in self._children.values())


class Embedding(Layer):
    """tf.keras.layers.Embedding(input_dim, output_dim, embeddings_initializer)."""

    def __init__(self, input_dim, output_dim, embeddings_initializer="random_uniform", input_length=None,
                 embeddings_regularizer=None, name=None, device=None):
        super().__init__(name, device)
        self.input_dim, self.output_dim = int(input_dim), int(output_dim)
        self.add_weight("embeddings", (self.input_dim, self.output_dim), embeddings_initializer)

    @property
    def table(self) -> torch.Tensor:
        return self._w["embeddings"]

    def call(self, ids, **kwargs):
        """ids (...,) int32/float32 -> (..., D): single-table gather through the fused kernel."""
        flat = ids.reshape(-1, 1)
        if flat.dtype not in (torch.int32, torch.float32):
            flat = flat.to(torch.int32)
        g = ops.TableGroup([self.table])
        out = ops.gather_concat(g, flat.contiguous())
        return out.view(*ids.shape, self.output_dim)


class Dense(Layer):
    """tf.keras.layers.Dense(units, activation, use_bias); built lazily on the first call.
    `activation` may be a Keras string, or a PReLU / Dice layer instance (src/ctr/din/model.py:52)."""

    def __init__(self, units, activation=None, use_bias=True, kernel_regularizer=None, name=None, device=None):
        super().__init__(name, device)
        self.units = int(units)
        self.use_bias = use_bias
        # a PReLU() / Dice() instance is tracked ONCE, under 'prelu' / 'dice' (build): bypass Layer.__setattr__, which
        # would register it a second time as child 'activation' (every weight twice in get_weights / named_weights)
        self.__dict__["activation"] = activation
        self.built = False

    def build(self, in_dim: int):
        self.add_weight("kernel", (in_dim, self.units), "glorot_uniform")
        if self.use_bias:
            self.add_weight("bias", (self.units,), "zeros")
        if isinstance(self.activation, PReLU):
            self.activation.build(self.units)
            self._children["prelu"] = self.activation
        elif isinstance(self.activation, Dice):
            self._children["dice"] = self.activation
        self.built = True

    def call(self, x, out=None, row_absmax=None, out_absmax=None, **kwargs):
        if not self.built:
            self.build(x.shape[-1])
        W, b = self._w["kernel"], self._w.get("bias")
        return self.apply(x, W, b, out=out, row_absmax=row_absmax, out_absmax=out_absmax)

    def apply(self, x, W, b, out=None, row_absmax=None, out_absmax=None):
        """row_absmax / out_absmax: the row maxima ops.dense's large-layer kernel scales by, handed from layer to layer
        (dense_chain) so that no layer re-reads its input to find them"""
        act = self.activation
        if isinstance(act, PReLU):
            return ops.dense(x, W, b, "prelu", act._w["alpha"], out=out, row_absmax=row_absmax, out_absmax=out_absmax)
        if isinstance(act, Dice):
            return act(ops.dense(x, W, b, None, out=out, row_absmax=row_absmax))     # Dice rescales: no maxima downstream
        return ops.dense(x, W, b, act, out=out, row_absmax=row_absmax, out_absmax=out_absmax)


def dense_chain(layers, x, out=None, first=None, row_absmax=None):
    """x -> layers[0] -> layers[1] -> ... (a tower's Dense stack).  Between two large layers the producer's epilogue delivers
    the row maxima the consumer's kernel scales by (csrc/dense_f16x2.hip).  first: optional (W, b) that replaces the first
    layer's own kernel / bias (a folded BatchNormalization, zero rows for pad columns).  row_absmax: the maxima of x's rows
    when the kernel that produced x delivered them."""
    n = len(layers)
    # layer i hands maxima to layer i + 1 when that one is large: rows >= 1024, K = layer i's width a multiple of 32
    wants = [i + 1 < n and x.dim() == 2 and x.shape[0] >= 1024 and layers[i].units % 32 == 0 and layers[i].units >= 64
             and layers[i + 1].units > 8 and not isinstance(layers[i].activation, Dice) for i in range(n)]
    pool = torch.zeros((sum(wants), x.shape[0]), dtype=torch.float32, device=x.device) if any(wants) else None   # one fill
    am, slot = row_absmax, 0
    for i, layer in enumerate(layers):
        o = out if i == n - 1 else None
        oam = None
        if wants[i]:
            oam, slot = pool[slot], slot + 1
        if i == 0 and first is not None:
            x = layer.apply(x, first[0], first[1], out=o, row_absmax=am, out_absmax=oam)
        else:
            x = layer(x, out=o, row_absmax=am, out_absmax=oam)
        am = oam
    return x


class PReLU(Layer):
    """tf.keras.layers.PReLU(): per-feature alpha, zero-initialised (== relu at init)."""

    def __init__(self, name=None, device=None):
        super().__init__(name, device)

    def build(self, units):
        if "alpha" not in self._w:
            self.add_weight("alpha", (units,), "zeros")


class BatchNormalization(Layer):
    """Inference-mode tf.keras.layers.BatchNormalization (eps 1e-3): y = (x-mean)*gamma/sqrt(var+eps)+beta.
    Exposes `fold(W, b)` so a following Dense absorbs it (no extra pass over the activations)."""

    def __init__(self, center=True, scale=True, epsilon=1e-3, trainable=True, name=None, device=None):
        super().__init__(name, device)
        self.center, self.scale, self.epsilon = center, scale, epsilon
        self.built = False

    def build(self, d):
        if self.scale:
            self.add_weight("gamma", (d,), "ones")
        if self.center:
            self.add_weight("beta", (d,), "zeros")
        self.add_weight("moving_mean", (d,), "zeros")
        self.add_weight("moving_variance", (d,), "ones")
        self.built = True

    def scale_shift(self, d):
        if not self.built:
            self.build(d)
        inv = torch.rsqrt(self._w["moving_variance"] + self.epsilon)
        if self.scale:
            inv = inv * self._w["gamma"]
        shift = -self._w["moving_mean"] * inv
        if self.center:
            shift = shift + self._w["beta"]
        return inv, shift

    def fold(self, W: torch.Tensor, b: Optional[torch.Tensor]):
        """(BN(x)) @ W + b == x @ (inv[:,None]*W) + (shift @ W + b)  — a parameter transform."""
        inv, shift = self.scale_shift(W.shape[0])
        Wf = (inv[:, None] * W).contiguous()
        bf = shift @ W
        if b is not None:
            bf = bf + b
        return Wf, bf.contiguous()


class Dice(Layer):
    """src/ctr/layers/modules.py:327-337: BN(center=False, scale=False) + scalar alpha."""

    def __init__(self, name=None, device=None):
        super().__init__(name, device)
        self.bn = BatchNormalization(center=False, scale=False, device=device)
        # reference: add_weight(shape=(), name='alpha') -> Keras' default glorot_uniform on fans (1,1);
        # the initial value is not parity-relevant (tests load explicit weights), zero here.
        self.add_weight("alpha", (), "zeros")

    def call(self, x, **kwargs):
        if not self.bn.built:
            self.bn.build(x.shape[-1])
        return ops.dice(x, self._w["alpha"].reshape(1), self.bn._w["moving_mean"], self.bn._w["moving_variance"],
                        self.bn.epsilon)


class Dropout(Layer):
    """Identity at inference."""

    def __init__(self, rate=0.0, name=None, device=None):
        super().__init__(name, device)
        self.rate = rate

    def call(self, x, **kwargs):
        return x


class LayerNormalization(Layer):
    def __init__(self, epsilon=1e-3, name=None, device=None):
        super().__init__(name, device)
        self.epsilon = epsilon
        self.built = False

    def build(self, d):
        self.add_weight("gamma", (d,), "ones")
        self.add_weight("beta", (d,), "zeros")
        self.built = True

    def call(self, x, residual=None, row_mask=None, **kwargs):
        if not self.built:
            self.build(x.shape[-1])
        return ops.layernorm_residual(x, residual, self._w["gamma"], self._w["beta"], self.epsilon, row_mask)


class Model(Layer):
    """tf.keras.Model-shaped base.  `build_graph()` of the reference wraps `call` in a functional
    Model used only for fit/predict; here it returns a `Graph` with the same forward."""

    def build_graph(self, **kwargs):
        return Graph(self)

    def summary(self, **kwargs):
        print(f"{type(self).__name__}: {self.count_params():,} parameters")
        return Graph(self)


class Graph:
    """Callable handle returned by build_graph(): graph(inputs) / graph.predict(inputs)."""

    def __init__(self, model: Model):
        self.model = model

    def __call__(self, inputs, **kwargs):
        return self.model.call(inputs, **kwargs)

    def predict(self, inputs, **kwargs):
        out = self.model.call(inputs, **kwargs)
        torch.cuda.synchronize()
        return out.cpu().numpy() if isinstance(out, torch.Tensor) else out

    def summary(self):
        print(f"{type(self.model).__name__}: {self.model.count_params():,} parameters")


def to_device_ids(x, device) -> torch.Tensor:
    """numpy / tensor ids -> GPU tensor, int32 or float32 preserved (Keras cast semantics live in
    the kernel)."""
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if x.dtype == torch.int64:
        # ids outside int32 must stay out of range after the narrowing (a wrapped 2**32 + 5 would alias row 5): they
        # saturate to -1 / INT32_MAX, which no table holds, so the kernels report them like any other bad id
        x = x.clamp(-1, 2 ** 31 - 1).to(torch.int32)
    elif x.dtype in (torch.int16, torch.int8, torch.uint8):
        x = x.to(torch.int32)
    elif x.dtype == torch.float64:
        x = x.to(torch.float32)
    return x.to(device)


def to_device_f32(x, device) -> torch.Tensor:
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    return x.to(device=device, dtype=torch.float32)
