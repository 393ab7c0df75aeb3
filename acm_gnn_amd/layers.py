"""``GraphConvolution`` / ``MLP`` with the reference's interface, running on the
MI355X kernels.

Mirror of ACM-Geometric/layers.py:13-163 and ACM-Pytorch/models/layers.py:14-285:
same constructor signature, parameter names (``state_dict`` compatible),
initialisation scheme, ``forward(input, adj_low, adj_high, adj_low_unnormalized)``
signature, ``att_low/att_high/att_mlp[/att_struc_vec_low]`` attributes after a
forward, and ``__repr__``.  The arithmetic is the fused HIP path in
``functional.AcmConvFunction``; nothing here falls back to torch ops.

One thing the reference decides implicitly is made explicit (SURVEY.md quirk
Q1): whether LayerNorm feeds the attention logits of ``acmgcnp``/``acmgcnpp``.
ACM-Geometric does (layers.py:59,67), ACM-Pytorch never does because its layer
tests for the spellings ``"acmgcn+"/"acmgcn++"`` (models/layers.py:96,123).
``attn_layernorm=None`` resolves to ``DEFAULT_ATTN_LAYERNORM`` (True, the
ACM-Geometric behaviour) -- ``acm_gnn_amd.dropin`` flips it for ACM-Pytorch.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.parameter import Parameter

from . import functional as AF
from . import tuning
from .graph import FilterOperators, SparseFeatures, operators_for

_TORCH_DROPOUT = F.dropout          # to notice a patched F.dropout (mask replay in tests): see _csr_input

DEFAULT_ATTN_LAYERNORM = True
_PLUS_LITERAL = ("acmgcn+", "acmgcn++")          # spellings for which ACM-Pytorch's LN fires
_PLUS = ("acmgcnp", "acmgcnpp") + _PLUS_LITERAL


def _default_device():
    # the reference places parameters on cuda:0 when a GPU exists (layers.py:10-11)
    return torch.device("cuda:0" if torch.cuda.is_available() else "cpu")


class GraphConvolution(nn.Module):
    def __init__(self, in_features, out_features, nnodes, model_type, output_layer=0, variant=False,
                 structure_info=0, attn_layernorm=None, gather_dtype=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.output_layer, self.model_type = output_layer, model_type
        self.structure_info, self.variant = structure_info, variant
        if attn_layernorm is None:
            attn_layernorm = True if model_type in _PLUS_LITERAL else DEFAULT_ATTN_LAYERNORM
        self.attn_layernorm = bool(attn_layernorm)
        # storage type of the gathered operand on the wide literal path: "fp32" (reference numerics) or "bf16"
        # (half the gather bytes, ~3 decimal digits on that operand; fp32 accumulation) -- BASELINE config 3
        self.gather_dtype = gather_dtype or "fp32"
        # evaluation passes over an unmodified input reuse P = A_low X of the previous pass (_eval_agg_holder); a plain
        # attribute so that a caller who edits features through views the version counter cannot see can switch it off
        self.eval_agg_cache = True
        self._att_raw, self._att_inv = None, None            # see the att_low / att_high / att_mlp properties
        dev = _default_device()

        def new(*shape):
            return Parameter(torch.empty(*shape, dtype=torch.float32, device=dev))

        self.weight_low, self.weight_high, self.weight_mlp = (new(in_features, out_features) for _ in range(3))
        self.att_vec_low, self.att_vec_high, self.att_vec_mlp = (new(out_features, 1) for _ in range(3))
        self.layer_norm_low, self.layer_norm_high, self.layer_norm_mlp = (nn.LayerNorm(out_features) for _ in range(3))
        self.layer_norm_struc_low, self.layer_norm_struc_high = nn.LayerNorm(out_features), nn.LayerNorm(out_features)
        self.att_struc_low = new(out_features, 1)
        self.struc_low = new(nnodes, out_features)
        k = 3 if structure_info == 0 else 4
        self.att_vec = new(k, k)
        self.reset_parameters()

    def reset_parameters(self):
        # same draw order as the reference (layers.py:31-54) so a seeded CPU init matches it
        bound_w = 1.0 / math.sqrt(self.weight_mlp.size(1))
        bound_v = 1.0 / math.sqrt(self.att_vec_mlp.size(1))
        bound_m = 1.0 / math.sqrt(self.att_vec.size(1))
        for p in (self.weight_low, self.weight_high, self.weight_mlp, self.struc_low):
            p.data.uniform_(-bound_w, bound_w)
        for p in (self.att_vec_high, self.att_vec_low, self.att_vec_mlp, self.att_struc_low):
            p.data.uniform_(-bound_v, bound_v)
        self.att_vec.data.uniform_(-bound_m, bound_m)
        for ln in (self.layer_norm_low, self.layer_norm_high, self.layer_norm_mlp,
                   self.layer_norm_struc_low, self.layer_norm_struc_high):
            ln.reset_parameters()

    # ------------------------------------------------------------------
    def _config(self):
        return AF.AcmConfig(self.model_type, self.variant, self.structure_info, self.attn_layernorm, self.gather_dtype)

    def _param_dict(self):
        # (straight from the modules' parameter tables: seventeen nn.Module.__getattr__ fall-backs per call were 15 % of the
        # layer's host time on the zero-edit route, which is entirely host-bound)
        p = self._parameters
        m = self._modules
        try:
            lo, hi, ml, sl = (m[k]._parameters for k in ("layer_norm_low", "layer_norm_high", "layer_norm_mlp", "layer_norm_struc_low"))
            return {
                "weight_low": p["weight_low"], "weight_high": p["weight_high"], "weight_mlp": p["weight_mlp"],
                "att_vec_low": p["att_vec_low"], "att_vec_high": p["att_vec_high"], "att_vec_mlp": p["att_vec_mlp"],
                "att_struc_low": p["att_struc_low"], "struc_low": p["struc_low"], "att_vec": p["att_vec"],
                "layer_norm_low.weight": lo["weight"], "layer_norm_low.bias": lo["bias"],
                "layer_norm_high.weight": hi["weight"], "layer_norm_high.bias": hi["bias"],
                "layer_norm_mlp.weight": ml["weight"], "layer_norm_mlp.bias": ml["bias"],
                "layer_norm_struc_low.weight": sl["weight"], "layer_norm_struc_low.bias": sl["bias"],
            }
        except KeyError:                              # (a subclass that keeps them elsewhere: the attribute route)
            pass
        return {
            "weight_low": self.weight_low, "weight_high": self.weight_high, "weight_mlp": self.weight_mlp,
            "att_vec_low": self.att_vec_low, "att_vec_high": self.att_vec_high, "att_vec_mlp": self.att_vec_mlp,
            "att_struc_low": self.att_struc_low, "struc_low": self.struc_low, "att_vec": self.att_vec,
            "layer_norm_low.weight": self.layer_norm_low.weight, "layer_norm_low.bias": self.layer_norm_low.bias,
            "layer_norm_high.weight": self.layer_norm_high.weight, "layer_norm_high.bias": self.layer_norm_high.bias,
            "layer_norm_mlp.weight": self.layer_norm_mlp.weight, "layer_norm_mlp.bias": self.layer_norm_mlp.bias,
            "layer_norm_struc_low.weight": self.layer_norm_struc_low.weight,
            "layer_norm_struc_low.bias": self.layer_norm_struc_low.bias,
        }

    # the drop-in route's CSR features (see _csr_input): inputs below this many elements are not support-checked per
    # step (the check's device-to-host copy would cost a small graph more than the dense projection does)
    CSR_CHECK_MIN_ELEMENTS = 1 << 24

    def _csr_input(self, x):
        """The reference's own model code in front of this layer (the drop-in route: its GCN applies ``F.dropout`` to the
        dense features and hands the result over, models.py:54) -> CSR features where that pays (tuning csr_features;
        graph.SparseFeatures.auto).  An evaluation pass sees the loader's feature tensor itself: its twin is made once and
        remembered as this layer's reference structure.  A training pass sees a fresh dropped copy every step: for large
        inputs it becomes that structure with the copy's values, after a check that the copy has no entry outside it
        (one read of x + one small host copy; Penn94-shaped first layer: 2 ms of dense projection saved); small inputs
        and anything that fails the check keep the dense projection."""
        if F.dropout is not _TORCH_DROPOUT:          # a mask-replay harness: the input stays what its masks were recorded for
            return x
        twin = SparseFeatures.known_twin(x)
        if twin is not None:
            return twin
        if not self.training:
            # (a layer that once refused a tensor of this shape does not count the nonzeros of every new tensor of that
            # shape -- wide hidden activations arrive as fresh objects each pass; the loader's tensor carries its answer)
            if getattr(x, "_acm_csr_twin", None) is None and getattr(self, "_csr_refused", None) == tuple(x.shape):
                return x
            out = SparseFeatures.auto(x)
            if out is not x:
                self._csr_ref = out
            else:
                self._csr_refused = tuple(x.shape)
            return out
        ref = getattr(self, "_csr_ref", None)
        if (ref is None or x.requires_grad or x.dtype != torch.float32 or x.dim() != 2
                or x.numel() < self.CSR_CHECK_MIN_ELEMENTS or tuning.HOST.csr_features <= 0
                or (x.is_cuda and torch.cuda.is_current_stream_capturing())):
            return x
        out = ref.twin_of_masked(x)
        return x if out is None else out

    def forward(self, input, adj_low, adj_high=None, adj_low_unnormalized=None, post_relu=False, post_scale=None,
                post_drop=None, rows_permuted=False, call=None, input_drop=None):
        """Reference signature plus optional keyword arguments: ``post_relu`` / ``post_scale`` fuse the
        caller's ``dropout(relu(out))`` (post_scale = keep-mask / (1 - p)) into the kernel epilogue;
        ``post_drop = (p, tag, functional.DropoutState)`` does the same with the mask generated in registers.
        ``input`` may carry zero columns beyond ``in_features`` (functional.dropout(..., pad_to=...)).
        ``rows_permuted``: the operators are relabelled (graph.relabel_by_degree) and the caller already works in
        that numbering -- input, post_scale and the result are rows of the relabelled graph (models.GCN keeps the
        hidden activations there); otherwise the layer translates at its boundary.
        ``input_drop = (p, tag, functional.DropoutState)``: the caller's dropout of ``input`` (models.py:54), applied by this
        layer -- inside the dense projection's tile loads where the shapes allow (functional.in_drop_supported), as a launch
        of its own otherwise.
        ``call``: the model call's functional.CallContext (models.GCN passes one per forward; default: a fresh one derived
        from the calling thread's ``with functional.deferred_reductions() / fused_loss_tail()`` blocks)."""
        mt = self.model_type
        if mt == "mlp":
            return AF.mm(input, self.weight_mlp)
        if mt in ("sgc", "gcn"):
            # the reference calls the dense torch.mm here, so adj_low must be dense (layers.py:83-85)
            if isinstance(adj_low, torch.Tensor) and adj_low.layout != torch.strided:
                raise RuntimeError("model_type 'sgc'/'gcn' multiplies with torch.mm: adj_low must be dense")
            return AF.mm(adj_low, AF.mm(input, self.weight_low))
        cfg = self._config()
        if isinstance(input, torch.Tensor) and input.layout != torch.strided:
            input = SparseFeatures.from_torch(input)          # torch-sparse features: CSR projection
        elif (isinstance(input, torch.Tensor) and input_drop is None and not rows_permuted
                and not (isinstance(adj_low, FilterOperators) and adj_low.sharded)):
            input = self._csr_input(input)                    # wide one-hot / bag-of-words features handed over dense
        if isinstance(adj_low, FilterOperators):
            ops = adj_low
        else:
            ops = operators_for(adj_low, adj_high, adj_low_unnormalized if cfg.n_channels == 4 else None)
        params = self._param_dict()
        raw_input = input
        translate = ops.perm is not None and not rows_permuted
        if ops.perm is not None:
            if cfg.n_channels == 4:                       # struc_low is a parameter in the caller's numbering
                params["struc_low"] = self.struc_low.index_select(0, ops.perm)
            if translate:
                input = input.permute_rows(ops.perm) if isinstance(input, SparseFeatures) else input.index_select(0, ops.perm)
                if post_scale is not None:
                    post_scale = post_scale.index_select(0, ops.perm)
        # an output layer without post-op may take a pending functional.fused_loss_tail request (row phase + loss + K3);
        # the request's labels / weights are rows of the numbering the layer works in
        tail_layer = (bool(self.output_layer) and not post_relu and post_scale is None and post_drop is None
                      and not translate)
        # P = A_low X of an earlier pass is reusable only for the SAME operand: with an input dropout in play -- carried by
        # the projection or applied right here -- the operand changes with the step counter while ``raw_input`` (the
        # holder's key) does not (a stale P when the layer drops the input itself)
        holder = self._eval_agg_holder(raw_input, ops) if (input_drop is None or input_drop[0] <= 0) else None
        if input_drop is not None and not AF.in_drop_supported(input, ops, cfg, self.in_features, self.out_features):
            p_in, tag_in, st_in = input_drop          # the projection cannot carry it: the dropped copy, as a launch of its own
            input = AF.dropout(input, p_in, st_in, tag=tag_in, row_offset=ops.row_offset)
            input_drop = None
        out, att = AF.acm_conv(input, params, ops, cfg, post_relu, post_scale, post_drop, call=call,
                               tail_layer=tail_layer, agg_holder=holder, in_drop=input_drop)
        if translate:
            out = out.index_select(0, ops.inv_perm)
        # the mixing weights stay where the kernel wrote them; the attributes translate rows when they are read
        d = self.__dict__                       # (plain attributes: not through Module.__setattr__'s parameter / buffer checks)
        d["_att_raw"], d["_att_inv"], d["_att_k"] = att, ops.inv_perm, cfg.n_channels
        return out

    def _eval_agg_holder(self, x, ops):
        """The holder through which an aggregate-first forward reuses P = A_low X of the previous pass over the SAME dense
        input -- same tensor object, unmodified since (``_version``), same operators, no gradient asked for it: every
        evaluation pass over a static feature matrix after the first, and (round 4) every TRAINING step of a model without
        input dropout (the reference's twitch-gamer ACM-GCN+ runs: dropout 0), whose first-layer gather -- a quarter of a
        step's gathered bytes -- then runs once per training run instead of once per step.  With input dropout the input
        is a fresh tensor every step and simply never matches."""
        if not isinstance(x, torch.Tensor) or x.requires_grad or x.grad_fn is not None or not self.eval_agg_cache:
            return None
        # the entry keeps the input alive, so its storage cannot be handed to another tensor while the entry exists; an
        # in-place edit bumps the version counter (shared by every alias of the storage)
        # (the entry also keeps the operators alive and compares them by identity: an id() of a collected object can be
        # handed to the next one)
        key = (x.data_ptr(), x._version, tuple(x.shape), tuple(x.stride()))
        # one entry per ROLE: a training step and the evaluation pass of the same epoch may hand over different tensor
        # objects (relabelled operators: the training loop permutes x once, an evaluation forward permutes it again) --
        # with a single entry each replaced the other's, P was recomputed every pass and a captured training step lost the
        # P its graph reads
        role = "train" if (self.training and torch.is_grad_enabled()) else "eval"
        slots = self.__dict__.setdefault("_eval_agg", {})
        cached = slots.get(role)
        if cached is None or cached[0] != key or cached[3] is not ops:
            cached = (key, x, {"agg": None}, ops)
            slots[role] = cached
        return cached[2]

    def held_entries(self):
        """The cache entries as they are now (a captured step keeps them: train.TrainStep / EvalStep ``_held``)."""
        return list(self.__dict__.get("_eval_agg", {}).values())

    # After a forward the reference's layer holds att_low / att_high / att_mlp (/ att_struc_vec_low): N x 1 tensors, 0
    # before the first call (layers.py:17, 91, 107).  Here they are views of the kernel's [n, 4] output, translated
    # back to the caller's node numbering on access when the operators are relabelled.
    def _att_col(self, c):
        if self._att_raw is None:
            return 0
        if self._att_inv is not None:
            self._att_raw, self._att_inv = self._att_raw.index_select(0, self._att_inv), None
        return self._att_raw[:, c:c + 1]

    att_low = property(lambda self: self._att_col(0))
    att_high = property(lambda self: self._att_col(1))
    att_mlp = property(lambda self: self._att_col(2))

    @property
    def att_struc_vec_low(self):
        if self._att_raw is None or self._att_k != 4:
            raise AttributeError("att_struc_vec_low exists after a forward with structure_info (layers.py:111)")
        return self._att_col(3)

    def __repr__(self):
        return f"{self.__class__.__name__} ({self.in_features} -> {self.out_features})"


class MLP(nn.Module):
    """Linear stack with ReLU -> BatchNorm -> dropout between layers (the residual
    branch of ACM-GCN++ uses num_layers=1, i.e. a single Linear; layers.py:123-163)."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout=0.5):
        super().__init__()
        self.lins, self.bns = nn.ModuleList(), nn.ModuleList()
        if num_layers == 1:
            self.lins.append(nn.Linear(in_channels, out_channels))
            self.bns.append(nn.BatchNorm1d(out_channels))
        else:
            self.lins.append(nn.Linear(in_channels, hidden_channels))
            self.bns.append(nn.BatchNorm1d(hidden_channels))
            for _ in range(num_layers - 2):
                self.lins.append(nn.Linear(hidden_channels, hidden_channels))
                self.bns.append(nn.BatchNorm1d(hidden_channels))
            self.lins.append(nn.Linear(hidden_channels, out_channels))
        self.dropout = dropout

    def reset_parameters(self):
        for m in list(self.lins) + list(self.bns):
            m.reset_parameters()

    def forward(self, data, input_tensor=False):
        x = data if input_tensor else data.graph["node_feat"]
        for lin, bn in zip(self.lins[:-1], self.bns):
            x = torch.relu(lin(x))
            x = bn(x)
            x = nn.functional.dropout(x, p=self.dropout, training=self.training)
        return self.lins[-1](x)
