"""Training side of the path: what `model.compile(...); model.fit(...)` of the reference's train
scripts does around the forward — src/ctr/deep_fm/train.py:44-68 (DeepFM), src/ctr/fm/train.py:43-67, and the same
recipe for DLRM / DCN (which ship without a train script).

  * a tape of backward closures over the HIP kernels (csrc/train_ops.hip + the forward kernels reused for the
    GEMM-shaped parts): Dense, training-mode BatchNormalization, the ctr DNN, embedding gather (+ the fused gather +
    pairwise dot), FM layer, cross network, sigmoid + Keras binary cross-entropy;
  * training-mode forwards of DLRM ('cat' / 'dot'), DeepFM and DCN that mirror the models' call() line by line
    (BatchNormalization uses BATCH statistics, as Keras does under fit());
  * Keras-Adam (TF2 defaults) with the models' l2 regularisers folded in: the reference's exact dense form, or the
    lazy row-wise form for embedding tables (a documented deviation, recamd.h rec_adam_rows_f32);
  * data-parallel replicas: `allreduce` merges the gradients before the optimiser step (MirroredStrategy, C1);
  * `Trainer`: compile / fit / evaluate / predict with BCE + AUC, validation_split, EarlyStopping(monitor='val_loss',
    patience, restore_best_weights) and weights-only checkpoints (the ModelCheckpoint the scripts keep commented out).

Nothing here computes on the CPU; there is no autograd engine — every gradient is an explicit kernel."""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import nn, ops
from ._lib import C


def _s():
    return torch.cuda.current_stream().cuda_stream


def _ws(nbytes, dev):
    return torch.empty(max(1, int(nbytes)), dtype=torch.uint8, device=dev)


# ---- thin wrappers of the backward kernels ---------------------------------------------------------------------
def transpose(x: torch.Tensor) -> torch.Tensor:
    M, N = x.shape
    out = torch.empty((N, M), dtype=torch.float32, device=x.device)
    C.transpose_f32(x.data_ptr(), M, N, x.stride(0), out.data_ptr(), _s())
    return out


def colsum(a: torch.Tensor, b: Optional[torch.Tensor] = None, row_w: Optional[torch.Tensor] = None) -> torch.Tensor:
    M, N = a.shape
    out = torch.empty(N, dtype=torch.float32, device=a.device)
    ws = _ws(C.colsum_workspace_bytes(M, N), a.device)
    C.colsum_f32(a.data_ptr(), a.stride(0), 0 if b is None else b.data_ptr(), 0 if b is None else b.stride(0),
                 0 if row_w is None else row_w.data_ptr(), M, N, out.data_ptr(), ws.data_ptr(), _s())
    return out


def weight_grad(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    """dW = x^T dy for x (M, K), dy (M, N).  Small kernels under a long batch axis (the projections of the attention
    models: M = batch x positions) take rec_wgrad_small_f32, which splits the rows over workgroups; everything else is the
    forward GEMM on the transposed operand."""
    M, K = x.shape
    N = dy.shape[1]
    if N <= 256 and K <= 64 * (256 // N) and K + N <= 384 and M >= 2048 and x.stride(1) == 1 and dy.stride(1) == 1:
        out = torch.empty((K, N), dtype=torch.float32, device=x.device)
        ws = _ws(C.wgrad_small_workspace_bytes(M, K, N), x.device)
        C.wgrad_small_f32(x.data_ptr(), x.stride(0), dy.data_ptr(), dy.stride(0), M, K, N, out.data_ptr(), ws.data_ptr(), _s())
        return out
    xc = x if x.is_contiguous() else x.contiguous()
    xt = transpose(xc)                                  # (K, M): the batch is the reduction axis of dW
    if M >= 2048 and ((K + 127) // 128) * ((N + 127) // 128) < 128 and dy.is_contiguous() and N > 8:
        # a handful of output tiles under a long reduction: split-K (rec_dense_splitk_f32) instead of a few workgroups
        # walking the whole batch
        out = torch.empty((K, N), dtype=torch.float32, device=x.device)
        ws = _ws(C.dense_splitk_workspace_bytes(K, M, N), x.device)
        C.dense_splitk_f32(xt.data_ptr(), xt.stride(0), dy.data_ptr(), K, M, N, out.data_ptr(), ws.data_ptr(), _s())
        return out
    return ops.dense(xt, dy)


def sum_squares(t: torch.Tensor) -> torch.Tensor:
    """sum(t^2) as a 1-element device tensor (regularisation losses)"""
    flat = t.reshape(-1)
    n = flat.numel()
    cols = 256 if n % 256 == 0 else (64 if n % 64 == 0 else 1)
    v = flat.view(n // cols, cols)
    part = colsum(v, v)
    return colsum(part.view(-1, 1))


class Tape:
    """Backward closures in forward order; `grads` collects parameter gradients by name."""

    def __init__(self, seed: int = 0):
        self.ops: List[Callable[[], None]] = []
        self.grads: Dict[str, torch.Tensor] = {}
        self.seed = seed          # Dropout masks are a function of (seed, position on the tape)

    def add_grad(self, name: str, g: torch.Tensor):
        if name in self.grads:
            self.grads[name] = ops.axpby_act(self.grads[name], g, 1.0, 1.0, None)
        else:
            self.grads[name] = g

    def backward(self):
        for fn in reversed(self.ops):
            fn()
        self.ops.clear()


class Var:
    """An activation with a gradient slot (filled by the consumers' backward closures)."""
    __slots__ = ("v", "g")

    def __init__(self, v: torch.Tensor):
        self.v, self.g = v, None

    def acc(self, g: torch.Tensor):
        self.g = g if self.g is None else ops.axpby_act(self.g, g, 1.0, 1.0, None)


# ---- training-mode layers ----------------------------------------------------------------------------------------
def dense_fwd(tape: Tape, layer: nn.Dense, name: str, x: Var) -> Var:
    """Dense(units, activation): y = act(x W + b); backward = act', column sums, two GEMMs on transposed operands.
    `activation` may be a PReLU() / Dice() layer instance (src/ctr/din/model.py:52): the pre-activation is kept and the
    activation runs as its own tape op (rec_prelu_* / rec_dice_train_* around a training-mode BatchNormalization)."""
    xin = x.v if x.v.stride(1) == 1 and x.v.dim() == 2 else x.v.contiguous()
    if not layer.built:
        layer.build(xin.shape[-1])
    act = layer.activation
    layer_act = act if isinstance(act, (nn.PReLU, nn.Dice)) else None
    if layer_act is not None:
        act = None
    elif act is not None and not isinstance(act, str):
        raise NotImplementedError(f"training: Dense with activation {act!r} has no backward here")
    W, b = layer._w["kernel"], layer._w.get("bias")
    y = Var(ops.dense(xin, W, b, act))

    def bwd():
        dy = y.g if y.g.is_contiguous() else y.g.contiguous()
        if act not in (None, "linear", "none"):
            dy = dy.clone()
            C.act_grad_f32(dy.data_ptr(), dy.stride(0), y.v.data_ptr(), y.v.stride(0), dy.shape[0], dy.shape[1],
                           ops._act_id(act), _s())
        if b is not None:
            tape.add_grad(name + "/bias", colsum(dy))
        tape.add_grad(name + "/kernel", weight_grad(xin, dy))                   # dW = X^T dY
        x.acc(ops.dense(dy, transpose(W)))                                      # dX = dY W^T
    tape.ops.append(bwd)
    if isinstance(layer_act, nn.PReLU):
        return prelu_fwd(tape, layer_act, name + "/prelu", y)
    if isinstance(layer_act, nn.Dice):
        return dice_fwd(tape, layer_act, name + "/dice", y)
    return y


def prelu_fwd(tape: Tape, prelu: nn.PReLU, name: str, z: Var) -> Var:
    """tf.keras.layers.PReLU(): y = z >= 0 ? z : alpha[n] z with a per-feature alpha"""
    zv = z.v if z.v.is_contiguous() else z.v.contiguous()
    M, N = zv.shape
    alpha = prelu._w["alpha"]
    y = Var(torch.empty_like(zv))
    C.prelu_f32(zv.data_ptr(), zv.stride(0), alpha.data_ptr(), M, N, y.v.data_ptr(), y.v.stride(0), _s())

    def bwd():
        dy = y.g if y.g.is_contiguous() else y.g.contiguous()
        dz, nz = torch.empty_like(zv), torch.empty_like(zv)
        C.prelu_grad_f32(zv.data_ptr(), zv.stride(0), alpha.data_ptr(), dy.data_ptr(), dy.stride(0), M, N, dz.data_ptr(),
                         nz.data_ptr(), _s())
        tape.add_grad(name + "/alpha", colsum(dy, nz))
        z.acc(dz)
    tape.ops.append(bwd)
    return y


def dice_fwd(tape: Tape, dice: nn.Dice, name: str, x: Var) -> Var:
    """Dice (src/ctr/layers/modules.py:333-337), training: p = sigmoid(BatchNormalization(center=False, scale=False)(x)) on
    BATCH statistics; y = alpha (1 - p) x + p x"""
    xv = x.v if x.v.is_contiguous() else x.v.contiguous()
    xn = bn_fwd(tape, dice.bn, name + "/bn", x)
    alpha = dice._w["alpha"].reshape(1)
    y = Var(torch.empty_like(xv))
    C.dice_train_f32(xv.data_ptr(), xn.v.data_ptr(), alpha.data_ptr(), xv.numel(), y.v.data_ptr(), _s())

    def bwd():
        dy = y.g if y.g.is_contiguous() else y.g.contiguous()
        dx, dxn, da = torch.empty_like(xv), torch.empty_like(xv), torch.empty_like(xv)
        C.dice_train_grad_f32(xv.data_ptr(), xn.v.data_ptr(), alpha.data_ptr(), dy.data_ptr(), xv.numel(), dx.data_ptr(),
                              dxn.data_ptr(), da.data_ptr(), _s())
        tape.add_grad(name + "/alpha", colsum(colsum(da).view(-1, 1)).reshape(()))
        x.acc(dx)
        xn.acc(dxn)              # reaches x through the BatchNormalization backward, which runs after this closure
    tape.ops.append(bwd)
    return y


def bn_fwd(tape: Tape, bn: nn.BatchNormalization, name: str, x: Var, momentum: float = 0.99) -> Var:
    """BatchNormalization(training=True): batch statistics + moving-average update (Keras defaults)."""
    xin = x.v
    M, N = xin.shape
    if not bn.built:
        bn.build(N)
    dev = xin.device
    y = torch.empty((M, N), dtype=torch.float32, device=dev)
    mean = torch.empty(N, dtype=torch.float32, device=dev)
    inv = torch.empty(N, dtype=torch.float32, device=dev)
    gamma, beta = bn._w.get("gamma"), bn._w.get("beta")
    ws = _ws(C.colsum_workspace_bytes(M, N), dev)
    C.bn_train_f32(xin.data_ptr(), xin.stride(0), M, N, ops._ptr(gamma), ops._ptr(beta), float(bn.epsilon), momentum,
                   bn._w["moving_mean"].data_ptr(), bn._w["moving_variance"].data_ptr(), y.data_ptr(), y.stride(0),
                   mean.data_ptr(), inv.data_ptr(), ws.data_ptr(), _s())
    ops.note_weights_written(bn._w["moving_mean"], bn._w["moving_variance"])
    out = Var(y)

    def bwd():
        dy = out.g
        dx = torch.empty((M, N), dtype=torch.float32, device=dev)
        dgamma = torch.empty(N, dtype=torch.float32, device=dev)
        dbeta = torch.empty(N, dtype=torch.float32, device=dev)
        w2 = _ws(C.bn_train_grad_workspace_bytes(M, N), dev)
        C.bn_train_grad_f32(xin.data_ptr(), xin.stride(0), dy.data_ptr(), dy.stride(0), M, N, ops._ptr(gamma),
                            mean.data_ptr(), inv.data_ptr(), dx.data_ptr(), dx.stride(0), dgamma.data_ptr(),
                            dbeta.data_ptr(), w2.data_ptr(), _s())
        if gamma is not None:
            tape.add_grad(name + "/gamma", dgamma)
        if beta is not None:
            tape.add_grad(name + "/beta", dbeta)
        x.acc(dx)
    tape.ops.append(bwd)
    return out


def dnn_fwd(tape: Tape, dnn, name: str, x: Var) -> Var:
    """ctr DNN (src/ctr/layers/modules.py:129-135): BatchNormalization()(x) -> Dense stack -> Dropout(rate)."""
    h = bn_fwd(tape, dnn.bn, name + "/bn", x)
    for i, layer in enumerate(dnn.dnn_network):
        h = dense_fwd(tape, layer, f"{name}/dense_{i}", h)
    rate = getattr(dnn.dropout, "rate", 0.0)
    if rate:
        from .train_attn import dropout_fwd
        h = dropout_fwd(tape, h, rate)
    return h


def _grad_group(model, group: ops.TableGroup, names: Sequence[str], state: "TrainState") -> ops.TableGroup:
    return ops.TableGroup([state.grad(n) for n in names], out_cols=group.out_cols)


def gather_concat_fwd(tape: Tape, state: "TrainState", group: ops.TableGroup, names: Sequence[str], ids: torch.Tensor,
                      out: Optional[torch.Tensor] = None) -> Var:
    y = Var(ops.gather_concat(group, ids, out=out))

    def bwd():
        ops.embedding_grad(_grad_group(None, group, names, state), ids, y.g)
    tape.ops.append(bwd)
    return y


concat_cols = ops.concat_cols


def gather_pairwise_dot_fwd(tape: Tape, state: "TrainState", group: ops.TableGroup, names: Sequence[str],
                            ids: torch.Tensor, dense: Var) -> Var:
    z = Var(ops.gather_pairwise_dot(group, ids, dense.v, append_dense=True))

    def bwd():
        dz = z.g if z.g.stride(1) == 1 else z.g.contiguous()
        B, D = dense.v.shape
        dd = torch.empty((B, D), dtype=torch.float32, device=dz.device)
        gg = _grad_group(None, group, names, state)
        C.gather_pairwise_dot_grad_f32(group.descs, gg.descs, ids.data_ptr(), ids.stride(0), dense.v.data_ptr(),
                                       dense.v.stride(0), B, dz.data_ptr(), dz.stride(0), 1, dd.data_ptr(), dd.stride(0), _s())
        dense.acc(dd)
    tape.ops.append(bwd)
    return z


def fm_fwd(tape: Tape, fm, name: str, first: Var, second: Var) -> Var:
    out = Var(ops.fm_layer(first.v, second.v, fm._w["w"]))

    def bwd():
        B, L1 = first.v.shape
        M = second.v.shape[1]
        dev = first.v.device
        dout = out.g.reshape(-1).contiguous()
        d_first = torch.empty((B, L1), dtype=torch.float32, device=dev)
        d_second = torch.empty((B, M), dtype=torch.float32, device=dev)
        dw = torch.empty(L1, dtype=torch.float32, device=dev)
        ws = _ws(C.fm_layer_grad_workspace_bytes(B, L1), dev)
        C.fm_layer_grad_f32(first.v.data_ptr(), first.v.stride(0), L1, second.v.data_ptr(), second.v.stride(0), M,
                            fm._w["w"].data_ptr(), dout.data_ptr(), B, d_first.data_ptr(), d_first.stride(0),
                            d_second.data_ptr(), d_second.stride(0), dw.data_ptr(), ws.data_ptr(), _s())
        tape.add_grad(name + "/w", dw.view(L1, 1))
        first.acc(d_first)
        second.acc(d_second)
    tape.ops.append(bwd)
    return out


def cross_fwd(tape: Tape, cross, name: str, x0: Var) -> Var:
    """CrossNetwork, training: the L layer outputs are kept (one literal-recurrence launch per layer)."""
    if not cross.built:
        cross.build(x0.v.shape[-1])
    W, Bv = cross._w["cross_weights"], cross._w["cross_bias"]
    L = W.shape[0]
    x0c = x0.v.contiguous()
    xs = [x0c]
    for l in range(L):  # x_{l+1} = x0 (x_l . w_l) + b_l + x_l  ==  one-layer cross of x_l with "x0" swapped in:
        # the one-layer kernel computes x_l (x_l . w) + b + x_l, so the recurrence is evaluated from its closed pieces
        s = ops.dense(xs[-1], W[l].reshape(-1, 1).contiguous())                      # (B,1) = x_l . w_l
        xs.append(ops.axpby_act(ops.scale_rows(x0c, s.reshape(-1)), ops.axpby_act(xs[-1], Bv[l][None, :].expand_as(x0c).contiguous(), 1.0, 1.0, None), 1.0, 1.0, None))
    out = Var(xs[-1])

    def bwd():
        g = out.g.contiguous().clone()
        B, dim = g.shape
        dx0 = torch.zeros((B, dim), dtype=torch.float32, device=g.device)
        dW = torch.empty((L, dim), dtype=torch.float32, device=g.device)
        dB = torch.empty((L, dim), dtype=torch.float32, device=g.device)
        ds = torch.empty(B, dtype=torch.float32, device=g.device)
        for l in reversed(range(L)):
            dB[l] = colsum(g)
            C.cross_layer_grad_f32(x0c.data_ptr(), xs[l].data_ptr(), W[l].contiguous().data_ptr(), dim, B, g.data_ptr(),
                                   dx0.data_ptr(), ds.data_ptr(), _s())
            dW[l] = colsum(xs[l], row_w=ds)
        tape.add_grad(name + "/cross_weights", dW)
        tape.add_grad(name + "/cross_bias", dB)
        x0.acc(ops.axpby_act(dx0, g, 1.0, 1.0, None))       # x_0 is also the first x_l
    tape.ops.append(bwd)
    return out


def sigmoid_bce(tape: Tape, logits: Sequence[Var], y_true: torch.Tensor, grad_scale: float = 1.0):
    """p = sigmoid(sum of the logit parts) (the models end in tf.nn.sigmoid(tf.add(...))), loss = mean Keras BCE.
    grad_scale = 1 / world under data parallelism: the all-reduce SUM of the replicas' gradients is then the gradient
    of the mean loss over the global batch."""
    p = ops.add_sigmoid(logits[0].v, logits[1].v if len(logits) > 1 else None)
    yt = y_true.reshape(-1).contiguous()
    loss = ops.binary_crossentropy(yt, p.reshape(-1))

    def bwd():
        n = yt.numel()
        dz = torch.empty(n, dtype=torch.float32, device=p.device)
        C.bce_sigmoid_grad_f32(yt.data_ptr(), p.reshape(-1).data_ptr(), n, grad_scale / n, dz.data_ptr(), _s())
        for part in logits:
            part.acc(dz.view(n, 1))
    tape.ops.append(bwd)
    return p, loss


# ---- parameters, gradients, optimiser ----------------------------------------------------------------------------
def named_weights(layer: nn.Layer, prefix: str = "") -> Dict[str, torch.Tensor]:
    out = {prefix + k: v for k, v in layer._w.items()}
    for cname, c in layer._children.items():
        out.update(named_weights(c, f"{prefix}{cname}/"))
    return out


class TrainState:
    """Gradient buffers of the embedding tables (dense, zero between steps) + optimiser slots."""

    def __init__(self, model: nn.Layer):
        self.model = model
        self._grads: Dict[str, torch.Tensor] = {}
        self._sharded: Dict[int, torch.Tensor] = {}
        self.owner_summed = set()      # row-sharded tables: gradients arrive summed at the owner, never all-reduced

    def grad(self, name: str) -> torch.Tensor:
        w = named_weights(self.model)[name]
        if name not in self._grads:
            self._grads[name] = torch.zeros_like(w)
        return self._grads[name]

    def sharded_grad(self, st, names: Sequence[str]) -> torch.Tensor:
        """Gradient arena of row-sharded tables (recamd.dist.ShardedTables `st`; names[f] = weight name of table f): one
        (F * rows_local, D) buffer in the arena's layout — what ShardedTables.backward scatter-adds into — whose
        per-table row ranges are the tables' gradient buffers.  These gradients are already summed over the ranks by the
        reverse all-to-all (every rank owns its rows): `owner_summed` keeps them out of the data-parallel all-reduce."""
        key = id(st)
        if key not in self._sharded:
            ga = torch.zeros_like(st.arena)
            self._sharded[key] = ga
            weights = named_weights(self.model)
            for f, n in enumerate(names):
                rows = weights[n].shape[0]
                self._grads[n] = ga[f * st.rows_local: f * st.rows_local + rows]
                self.owner_summed.add(n)
        return self._sharded[key]


class Adam:
    """tf.keras.optimizers.Adam (TF2 defaults lr 1e-3, b1 .9, b2 .999, eps 1e-7, amsgrad off) over every trainable
    weight of a model.  `l2` maps a weight name (or name prefix) to its regulariser coefficient c: the gradient 2 c w
    is fused into the update (embeddings_regularizer=l2(embed_reg), the FM layer's l2(w_reg)).  sparse_embeddings=True
    switches the embedding tables to the lazy row-wise update (rec_adam_rows_f32; deviation: untouched rows are not
    decayed) — the exact form touches every row of every table each step, 28 B per parameter."""

    def __init__(self, model: nn.Layer, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7, l2: Optional[Dict[str, float]] = None,
                 sparse_embeddings: bool = False, decay: float = 0.0):
        self.model, self.lr0, self.b1, self.b2, self.eps = model, learning_rate, beta_1, beta_2, epsilon
        self.lr = learning_rate
        self.decay = decay        # Adam(learning_rate, decay=...) of src/ctr/esmm/train.py:95: lr / (1 + decay * iterations)
        self.l2 = dict(l2 or {})
        self.sparse = sparse_embeddings
        self.step_no = 0
        self.m: Dict[str, torch.Tensor] = {}
        self.v: Dict[str, torch.Tensor] = {}
        self.stamp: Dict[str, torch.Tensor] = {}

    def _l2_of(self, name: str) -> float:
        if name in self.l2:
            return self.l2[name]
        best = 0.0
        for k, c in self.l2.items():
            if len(k) > 1 and (name.startswith(k) or name.endswith(k)):
                best = c
        return best

    def trainable(self) -> Dict[str, torch.Tensor]:
        return {k: v for k, v in named_weights(self.model).items()
                if not (k.endswith("moving_mean") or k.endswith("moving_variance"))}

    def apply(self, grads: Dict[str, torch.Tensor], state: TrainState, sparse_ids=None):
        """grads: dense-parameter gradients by name; embedding gradients come from `state`.  sparse_ids: list of
        (names, ids (B,F) int32) for the lazy update."""
        self.lr = self.lr0 / (1.0 + self.decay * self.step_no)      # `iterations` = steps taken so far
        self.step_no += 1
        weights = self.trainable()
        lazy = set()
        if self.sparse and sparse_ids:
            for names, ids in sparse_ids:
                lazy.update(names)
                tabs = [weights[n] for n in names]
                for n, t in zip(names, tabs):
                    if n not in self.m:
                        self.m[n], self.v[n] = torch.zeros_like(t), torch.zeros_like(t)
                        self.stamp[n] = torch.zeros((t.shape[0], 1), dtype=torch.int32, device=t.device)
                d = lambda ts: [(t.data_ptr(), int(t.shape[0]), int(t.shape[1]), 0) for t in ts]  # noqa: E731
                l2 = self._l2_of(names[0])
                C.adam_rows_f32(d(tabs), d([self.m[n] for n in names]), d([self.v[n] for n in names]),
                                d([state.grad(n) for n in names]),
                                [(self.stamp[n].data_ptr(), int(self.stamp[n].shape[0]), 1, 0) for n in names],
                                ids.data_ptr(), ids.stride(0), ids.shape[0], self.lr, self.b1, self.b2, self.eps,
                                self.step_no, l2, _s())
                ops.note_weights_written(*tabs)
        for name, w in weights.items():
            if name in lazy:
                continue
            if name in grads:
                g = grads[name].reshape(w.shape).contiguous()
            elif name in state._grads:
                g = state._grads[name]
            else:
                continue   # a weight the loss does not reach (e.g. an unused embedding)
            if name not in self.m:
                self.m[name], self.v[name] = torch.zeros_like(w), torch.zeros_like(w)
            ops.adam_step(w, self.m[name], self.v[name], g, self.step_no, self.lr, self.b1, self.b2, self.eps,
                          self._l2_of(name))
            if name in state._grads:
                state._grads[name].zero_()


# ---- training-mode forwards of the models ------------------------------------------------------------------------
def _embed_names(n: int) -> List[str]:
    return [f"embed_{i}/embeddings" for i in range(n)]


def dlrm_train_forward(tape: Tape, state: TrainState, m, inputs, y_true, grad_scale: float = 1.0):
    """src/ctr/dlrm/model.py:42-54 (intended form; 'dot' = the cited paper's interaction), training mode."""
    dense_inputs, sparse_inputs = inputs
    dense_inputs = nn.to_device_f32(dense_inputs, m.device)
    ids = nn.to_device_ids(sparse_inputs, m.device)
    names = _embed_names(len(m._group))
    dense_fea = dnn_fwd(tape, m.bot_dnn, "bot_dnn", Var(dense_inputs))
    if m.interaction == "dot":
        x = gather_pairwise_dot_fwd(tape, state, m._group, names, ids, dense_fea)
    else:
        # tf.concat([sparse_embed, dense_fea]) (:48): the gather writes its columns of the concat buffer in place
        W = m._group.width
        buf = torch.empty((ids.shape[0], W + dense_fea.v.shape[1]), dtype=torch.float32, device=m.device)
        ops.gather_concat(m._group, ids, out=buf)
        ops.copy_cols(dense_fea.v, buf[:, W:])
        x = Var(buf)
        emb = Var(buf[:, :W])

        def bwd_gather():
            if emb.g is not None:
                ops.embedding_grad(_grad_group(None, m._group, names, state), ids, emb.g)
        tape.ops.append(bwd_gather)

        def bwd():
            emb.acc(x.g[:, :W].contiguous())
            dense_fea.acc(x.g[:, W:].contiguous())
        tape.ops.append(bwd)
    top = dense_fwd(tape, m.final_dense, "final_dense", dnn_fwd(tape, m.top_dnn, "top_dnn", x))
    return sigmoid_bce(tape, [top], y_true, grad_scale)


def deepfm_train_forward(tape: Tape, state: TrainState, m, inputs, y_true, grad_scale: float = 1.0):
    """src/ctr/deep_fm/model.py:50-65, training mode."""
    dense_inputs, sparse_inputs = inputs
    dense_inputs = nn.to_device_f32(dense_inputs, m.device)
    ids = nn.to_device_ids(sparse_inputs, m.device)
    names = _embed_names(len(m._group))
    plain = ops.TableGroup(m._group.tables)                                 # tf.concat offsets 0, D, 2D, ...
    sparse_embed = gather_concat_fwd(tape, state, plain, names, ids)        # :53
    embeds = Var(concat_cols([dense_inputs, sparse_embed.v]))              # :56
    nd = dense_inputs.shape[1]

    def bwd_cat():
        sparse_embed.acc(embeds.g[:, nd:].contiguous())
    tape.ops.append(bwd_cat)
    fm_out = fm_fwd(tape, m.fm, "fm", embeds, sparse_embed)                 # :59
    deep = dense_fwd(tape, m.dense, "dense", dnn_fwd(tape, m.dnn, "dnn", embeds))   # :61-62
    return sigmoid_bce(tape, [fm_out, deep], y_true, grad_scale)            # :64


def dcn_train_forward(tape: Tape, state: TrainState, m, inputs, y_true, grad_scale: float = 1.0):
    """src/ctr/dcn/model.py:45-57, training mode."""
    ids = nn.to_device_ids(inputs, m.device)
    names = _embed_names(len(m._group))
    x = gather_concat_fwd(tape, state, ops.TableGroup(m._group.tables), names, ids)      # :47
    cross_x = cross_fwd(tape, m.cross_network, "cross_network", x)                        # :51
    dnn_x = dnn_fwd(tape, m.dnn_network, "dnn_network", x)                                # :53
    total = Var(concat_cols([cross_x.v, dnn_x.v]))                                       # :55
    wc = cross_x.v.shape[1]

    def bwd():
        cross_x.acc(total.g[:, :wc].contiguous())
        dnn_x.acc(total.g[:, wc:].contiguous())
    tape.ops.append(bwd)
    out = dense_fwd(tape, m.dense_final, "dense_final", total)                            # :56
    return sigmoid_bce(tape, [out], y_true, grad_scale)


TRAIN_FORWARDS = {"DLRM": dlrm_train_forward, "DeepFM": deepfm_train_forward, "DCN": dcn_train_forward}


def train_forward_of(model):
    """the training-mode forward registered for the model's class: by module-qualified name where two mirrors share a class
    name (ctr FM / match FM), else by the bare class name; None if there is none"""
    cls = type(model)
    return TRAIN_FORWARDS.get(f"{cls.__module__}.{cls.__name__}") or TRAIN_FORWARDS.get(cls.__name__)


def default_l2(model) -> Dict[str, float]:
    """the regularisers the reference models attach: embeddings_regularizer=l2(embed_reg) on every table
    (e.g. src/ctr/dlrm/model.py:35, deep_fm/model.py:36, sasrec/model.py:44), the FM layer's l2(w_reg)
    (src/ctr/layers/modules.py:52-55), classic FM's l2(w_reg) / l2(v_reg) (src/ctr/fm/model.py:27,31) and the
    kernel_regularizer=l2(1e-4) default of the ctr MultiHeadAttention's projections (src/ctr/layers/modules.py:179,258-268,320)"""
    l2 = {}
    reg = getattr(model, "embed_reg", None)
    if reg:
        l2["/embeddings"] = float(reg)
    fm = getattr(model, "fm", None)
    if fm is not None and getattr(fm, "w_reg", 0):
        l2["fm/w"] = float(fm.w_reg)
    cross = getattr(model, "cross_network", None)
    if cross is not None:
        if getattr(cross, "reg_w", 0):
            l2["cross_network/cross_weights"] = float(cross.reg_w)
        if getattr(cross, "reg_b", 0):
            l2["cross_network/cross_bias"] = float(cross.reg_b)
    if type(model).__name__ == "FM":
        if getattr(model, "w_reg", 0):
            l2["w"] = float(model.w_reg)
        if getattr(model, "v_reg", 0):
            l2["V"] = float(model.v_reg)
    for i, layer in enumerate(getattr(model, "attention_layers", []) or []):
        c = getattr(layer, "_l2_reg", None)
        if c:
            for n in ("Wq", "Wk", "Wv", "W0"):
                l2[f"attention_{i}/{n}"] = float(c)
    return l2


def compute_gradients(model, state: TrainState, inputs, y_true, grad_scale: float = 1.0, seed: int = 0):
    """training-mode forward + backward of one batch: (predictions, loss, {name: dense-parameter gradient}); the
    embedding-table gradients are scatter-added into `state`.  loss = mean BCE against y_true, or the model's own
    add_loss when it has no labels (SASRec: y_true = None, predictions = the logits).  `seed` drives the Dropout masks."""
    from . import train_attn  # noqa: F401  (registers the other mirrors' forwards)
    fwd = train_forward_of(model)
    tape = Tape(seed)
    if isinstance(y_true, (list, tuple)):          # several targets (ESMM: [ctr, cvr])
        y = [nn.to_device_f32(np.asarray(t, np.float32), model.device) for t in y_true]
    else:
        y = None if y_true is None else nn.to_device_f32(y_true, model.device)
    p, loss = fwd(tape, state, model, inputs, y, grad_scale)
    tape.backward()
    return p, loss, tape.grads


def train_step(model, opt: Adam, state: TrainState, inputs, y_true, allreduce: Optional[Callable] = None,
               world: int = 1):
    """One optimiser step: training-mode forward, backward, [gradient merge over the replicas], Adam.  Returns
    (predictions, mean BCE of this replica's batch) as device tensors.

    allreduce(t): in-place SUM over the data-parallel replicas (recamd.dist.ShardedTables.allreduce_sum_ /
    Comm.allreduce_sum_).  Each replica scales its loss gradient by 1 / world, so the summed gradient is that of the
    mean loss over the GLOBAL batch — what MirroredStrategy does (src/ctr/fm/train.py:43-45); BatchNormalization
    keeps per-replica batch statistics, as there."""
    dp = allreduce is not None and world > 1
    if dp and opt.sparse:
        # the lazy row-wise update touches (and clears) only the rows of THIS replica's ids; after the all-reduce the
        # gradient buffers also hold the other replicas' rows, which would be neither applied nor cleared
        raise NotImplementedError("Adam(sparse_embeddings=True) under data parallelism: the lazy row-wise update needs "
                                  "the union of all replicas' ids; use the exact dense form (sparse_embeddings=False)")
    p, loss, grads = compute_gradients(model, state, inputs, y_true, 1.0 / world if dp else 1.0, seed=opt.step_no + 1)
    if dp:  # same order on every replica
        for k in sorted(grads):
            grads[k] = grads[k].contiguous()
            allreduce(grads[k])
        for k in sorted(state._grads):
            if k not in state.owner_summed:
                allreduce(state._grads[k])
    sparse_ids = None
    if opt.sparse:
        if type(model).__name__ not in ("DLRM", "DeepFM", "DCN"):
            raise NotImplementedError("Adam(sparse_embeddings=True) is wired for the (B, F) id matrix of DLRM / DeepFM / DCN")
        ids = inputs[1] if isinstance(inputs, (list, tuple)) else inputs
        ids = nn.to_device_ids(ids, model.device)
        sparse_ids = [(_embed_names(ids.shape[1]), ids)]
    opt.apply(grads, state, sparse_ids)
    return p, loss


# ---- compile / fit / evaluate -------------------------------------------------------------------------------------
class EarlyStopping:
    """tf.keras.callbacks.EarlyStopping(monitor='val_loss', patience, restore_best_weights) as the train scripts use it
    (src/ctr/deep_fm/train.py:62)."""

    def __init__(self, monitor="val_loss", patience=1, restore_best_weights=True, min_delta=0.0):
        self.monitor, self.patience, self.restore, self.min_delta = monitor, patience, restore_best_weights, min_delta
        self.best, self.wait, self.best_weights, self.stopped_epoch = None, 0, None, None

    def on_epoch_end(self, epoch: int, logs: Dict[str, float], trainer: "Trainer") -> bool:
        cur = logs.get(self.monitor)
        if cur is None:
            return False
        if self.best is None or cur < self.best - self.min_delta:
            self.best, self.wait = cur, 0
            if self.restore:
                self.best_weights = {k: v.clone() for k, v in named_weights(trainer.model).items()}
            return False
        self.wait += 1
        if self.wait >= self.patience:
            self.stopped_epoch = epoch
            if self.restore and self.best_weights is not None:
                trainer.load_state(self.best_weights)
            return True
        return False


class Trainer:
    """model.compile(loss=binary_crossentropy, optimizer=Adam(lr), metrics=[AUC()]); model.fit(...); model.evaluate(...)
    for the mirrors that have a training-mode forward (DLRM, DeepFM, DCN here; classic FM, AutoInt, DIN, SASRec in
    recamd/train_attn.py)."""

    def __init__(self, model):
        from . import train_attn  # noqa: F401  (registers the FM / AutoInt / DIN / SASRec forwards)
        if train_forward_of(model) is None:
            why = ""
            if type(model).__name__ == "YoutubeDNN":
                why = (": src/match/youtube_dnn/train.py minimises SampledSoftmaxLayer's tf.nn.sampled_softmax_loss, whose "
                       "log-uniform sampler is unseeded (src/match/layers/modules.py:43-61) - no reproducible objective to mirror")
            raise NotImplementedError(f"no training-mode forward for {type(model).__name__}{why}")
        self.model, self.opt, self.state = model, None, TrainState(model)
        self.allreduce, self.world = None, 1

    def compile(self, optimizer: Optional[Adam] = None, learning_rate: float = 1e-3, sparse_embeddings: bool = False,
                allreduce: Optional[Callable] = None, world: int = 1):
        if isinstance(optimizer, str):
            if optimizer.lower() != "adam":
                raise NotImplementedError(f"optimizer {optimizer!r}: the reference's train scripts all use Adam")
            optimizer = None
        self.opt = optimizer or Adam(self.model, learning_rate, l2=default_l2(self.model), sparse_embeddings=sparse_embeddings)
        if self.opt.sparse and allreduce is not None and world > 1:
            raise NotImplementedError("Adam(sparse_embeddings=True) is per-replica: not available with allreduce / world > 1")
        self.allreduce, self.world = allreduce, world
        return self

    # -- helpers
    @staticmethod
    def _slice(x, idx):
        """rows idx of the inputs: an array, a list of arrays, or (the match models) dicts {feature: (n, 1)} inside a list"""
        if isinstance(x, dict):
            return {k: np.asarray(v)[idx] for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [Trainer._slice(a, idx) for a in x]
        return np.asarray(x)[idx]

    @staticmethod
    def _len(x):
        if isinstance(x, dict):
            return len(next(iter(x.values())))
        return Trainer._len(x[0]) if isinstance(x, (list, tuple)) else len(x)

    def reg_loss_device(self) -> torch.Tensor:
        """sum of the l2 regularisation losses Keras adds to the reported loss — c * sum(w^2) per regularised weight — as a
        1-element DEVICE tensor (no host synchronisation: fit() accumulates it on the device)"""
        tot = torch.zeros(1, dtype=torch.float32, device=self.model.device)
        for name, w in named_weights(self.model).items():
            c = self.opt._l2_of(name)
            if c:
                tot = ops.axpby_act(tot, sum_squares(w), 1.0, float(c), None)
        return tot

    def reg_loss(self) -> float:
        return float(self.reg_loss_device().item())

    def predict(self, x, batch_size: int = 4096) -> np.ndarray:
        n = self._len(x)
        outs = []
        for lo in range(0, n, batch_size):
            idx = slice(lo, min(n, lo + batch_size))
            outs.append(self.model(self._slice(x, idx)).reshape(-1))
        return torch.cat(outs).cpu().numpy()

    def evaluate(self, x, y=None, batch_size: int = 4096):
        """[loss (BCE + regularisation losses, as Keras reports it), AUC] with INFERENCE-mode BatchNormalization / Dropout;
        y = None: [sample-weighted mean of the model's add_loss over the batches + regularisation losses]"""
        if y is None:
            n, tot = self._len(x), 0.0
            for lo in range(0, n, batch_size):
                idx = slice(lo, min(n, lo + batch_size))
                out = self.model(self._slice(x, idx))
                # the model's add_loss, or — a script whose loss is mean(y_pred) (src/match/utils/loss_util.py:11-13) — its output
                own = self.model.losses[-1] if getattr(self.model, "losses", None) else out.reshape(-1).mean()
                tot += float(own.item()) * (idx.stop - idx.start)
            return [tot / n + self.reg_loss()]
        p = torch.from_numpy(self.predict(x, batch_size)).to(self.model.device)
        yt = nn.to_device_f32(np.asarray(y, np.float32).reshape(-1), self.model.device)
        loss = float(ops.binary_crossentropy(yt, p).item()) + self.reg_loss()
        return [loss, float(ops.auc(yt, p).item())]

    def fit(self, x, y=None, batch_size: int = 32, epochs: int = 1, validation_split: float = 0.0, callbacks=(), shuffle=True,
            seed: int = 0, verbose: int = 0):
        """Keras semantics: the validation set is the LAST `validation_split` fraction (taken before shuffling); the
        epoch's `loss` is the sample-weighted mean of the batch losses (+ the regularisation losses at the end of each
        batch), `auc` is accumulated over the epoch's training predictions.  Shuffling uses numpy's
        default_rng(seed + epoch).permutation (Keras' own shuffle is unseeded: not reproducible there either).
        y = None: a model whose loss is its own add_loss (SASRec, src/match/sasrec/model.py:93-95) — `loss` only."""
        if getattr(train_forward_of(self.model), "ignores_labels", False):
            y = None                   # loss = mean(y_pred): Keras hands the labels to a loss function that drops them
        n = self._len(x)
        n_tr = int(math.floor(n * (1.0 - validation_split)))     # Keras: split_at = floor(n * (1 - validation_split))
        n_val = n - n_tr
        has_y = y is not None
        y_all = np.asarray(y, np.float32).reshape(-1) if has_y else None
        xt, xv = self._slice(x, slice(0, n_tr)), self._slice(x, slice(n_tr, n))
        yt_all, yv = (y_all[:n_tr], y_all[n_tr:]) if has_y else (None, None)
        history = {"loss": [], "auc": []} if has_y else {"loss": []}
        if n_val:
            history.update(dict(val_loss=[], val_auc=[]) if has_y else dict(val_loss=[]))
        for epoch in range(epochs):
            order = np.random.default_rng(seed + epoch).permutation(n_tr) if shuffle else np.arange(n_tr)
            # the epoch's loss is accumulated ON THE DEVICE (one host synchronisation per epoch, not several per batch)
            loss_sum, preds, labels = torch.zeros(1, dtype=torch.float32, device=self.model.device), [], []
            for lo in range(0, n_tr, batch_size):
                idx = order[lo:lo + batch_size]
                reg = self.reg_loss_device()   # Keras adds the regularisation losses of the weights the batch SAW
                p, loss = train_step(self.model, self.opt, self.state, self._slice(xt, idx), yt_all[idx] if has_y else None,
                                     self.allreduce, self.world)
                batch_loss = ops.axpby_act(loss.reshape(1), reg, 1.0, 1.0, None)
                loss_sum = ops.axpby_act(loss_sum, batch_loss, 1.0, float(len(idx)), None)
                if has_y:
                    preds.append(p.reshape(-1))
                    labels.append(yt_all[idx])
            logs = {"loss": float(loss_sum.item()) / n_tr}
            if has_y:
                lt = nn.to_device_f32(np.concatenate(labels), self.model.device)
                logs["auc"] = float(ops.auc(lt, torch.cat(preds)).item())
            if n_val:
                if has_y:
                    vl, va = self.evaluate(xv, yv, batch_size)
                    logs.update(val_loss=vl, val_auc=va)
                else:
                    logs.update(val_loss=self.evaluate(xv, None, batch_size)[0])
            for k, v in logs.items():
                history[k].append(v)
            if verbose:
                print(f"epoch {epoch + 1}/{epochs} " + " ".join(f"{k}={v:.6f}" for k, v in logs.items()), flush=True)
            if any(cb.on_epoch_end(epoch, logs, self) for cb in callbacks):
                break
        return history

    # -- weights-only checkpoints (ModelCheckpoint(save_weights_only=True) of src/ctr/fm/train.py:53-55)
    def save_weights(self, path: str) -> None:
        np.savez(path, **{k.replace("/", "|"): v.detach().cpu().numpy() for k, v in named_weights(self.model).items()})

    def load_weights(self, path: str) -> None:
        with np.load(path, allow_pickle=False) as z:
            self.model.set_weights({k.replace("|", "/"): z[k] for k in z.files})

    def load_state(self, weights: Dict[str, torch.Tensor]) -> None:
        cur = named_weights(self.model)
        for k, v in weights.items():
            cur[k].copy_(v)
        ops.note_weights_written(*cur.values())
