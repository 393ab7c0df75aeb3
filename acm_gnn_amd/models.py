"""Two-layer ACM model: the caller of the hot path (ACM-Geometric/models.py:23-76,
ACM-Pytorch/models/models.py:25-166), restated so the op can be trained and
measured on the GPU box (the reference's Python never travels there).

    x -> dropout -> GraphConvolution(nfeat -> nhid) -> relu -> dropout
      [-> + dropout(relu(Linear(x)))   for acmgcnpp]
      -> GraphConvolution(nhid -> nclass)

Same constructor signature and ``forward(x, adj_low, adj_high, adj_low_unnormalized)``.
Differences, on purpose: ``acmsgc`` and ``acmsnowball`` are constructible -- the reference's
constructor omits the layer's positional ``nnodes`` for both and raises TypeError (SURVEY.md quirk Q2;
ACM-Geometric/models.py:35,38-39).  With that argument supplied, ``acmsgc`` is a single linear ACM
layer nfeat -> nclass (its forward returns that layer's output: the reference's ``fea2`` is never
assigned on this path) and ``acmsnowball`` is the dense stack the reference's forward spells out
(models.py:57-64): layer k reads [x | h_0 | ... | h_{k-1}], the classifier layer reads all of them.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as AF
from .graph import FilterOperators, SparseFeatures
from .layers import GraphConvolution, MLP

_TORCH_DROPOUT = F.dropout          # to notice a patched F.dropout (mask replay in tests): see GCN.auto_csr

_TWO_LAYER = ("acmgcn", "acmgcnp", "acmgcnpp")


class GCN(nn.Module):
    def __init__(self, nfeat, nhid, nclass, nlayers, nnodes, dropout, model_type, structure_info,
                 variant=False, init_layers_X=1, attn_layernorm=None, gather_dtype=None):
        super().__init__()
        self.model_type, self.structure_info = model_type, structure_info
        self.nlayers, self.nnodes, self.dropout = nlayers, nnodes, dropout
        if model_type == "acmgcnpp":
            self.mlpX = MLP(nfeat, nhid, nhid, num_layers=init_layers_X, dropout=0)
        self.gcns, self.mlps = nn.ModuleList(), nn.ModuleList()
        kw = dict(model_type=model_type, variant=variant, structure_info=structure_info,
                  attn_layernorm=attn_layernorm, gather_dtype=gather_dtype)
        if model_type in _TWO_LAYER:
            self.gcns.append(GraphConvolution(nfeat, nhid, nnodes, **kw))
            self.gcns.append(GraphConvolution(nhid, nclass, nnodes, output_layer=1, **kw))
        elif model_type == "acmsgc":
            self.gcns.append(GraphConvolution(nfeat, nclass, nnodes, model_type=model_type, output_layer=1))
        elif model_type == "acmsnowball":
            # models.py:38-39 with the missing nnodes supplied; structure_info is not forwarded there either
            for k in range(nlayers):
                self.gcns.append(GraphConvolution(k * nhid + nfeat, nhid, nnodes, model_type=model_type, variant=variant,
                                                  attn_layernorm=attn_layernorm, gather_dtype=gather_dtype))
            self.gcns.append(GraphConvolution(nlayers * nhid + nfeat, nclass, nnodes, model_type=model_type,
                                              variant=variant, attn_layernorm=attn_layernorm, gather_dtype=gather_dtype))
        else:
            raise ValueError(f"GCN: unsupported model_type {model_type!r} "
                             "(acmgcn | acmgcnp | acmgcnpp | acmsgc | acmsnowball)")
        # The reference also registers two never-initialised 1x1 parameters (fea_param,
        # xX_param; models.py:41) that take no part in the forward.  They are kept so
        # state_dict keys and optimizer parameter lists line up, zero-filled.
        dev = self.gcns[0].weight_low.device
        self.fea_param = nn.Parameter(torch.zeros(1, 1, device=dev))
        self.xX_param = nn.Parameter(torch.zeros(1, 1, device=dev))
        # Counter-based dropout (functional.DropoutState): off by default -- F.dropout, like the reference.  A
        # training loop that advances ``dropout_state`` once per optimizer step (train.TrainStep does) may set
        # ``fused_dropout = True``: the masks are then generated inside the layer kernels.
        self.fused_dropout = False
        self.dropout_state = None
        self.reset_parameters()

    def _ones_like_hidden(self, n, f, device):
        key = (n, f, str(device))
        if getattr(self, "_ones_key", None) != key:
            self._ones_key, self._ones = key, torch.ones(n, f, device=device)
        return self._ones

    def reset_parameters(self):
        if self.model_type == "acmgcnpp":
            self.mlpX.reset_parameters()

    def _residual(self, x, adj_low, drop=None, call=None, pipe=None):
        """relu(Linear(x)) of the ACM-GCN++ branch (ACM-Geometric/models.py:26-27,55-56), optionally with the
        counter-based dropout in the same epilogue (``drop`` = (p, tag, state, row_offset)): one GEMM launch
        (functional.residual_linear; CSR features: acm_spmm_v + acm_bias_act).  Row-sharded: the Linear's weight / bias
        gradients are summed over the ranks.  mlpX stacks deeper than one Linear (init_layers_X > 1: BatchNorm between
        the layers) and Linears wider than 256 outputs (acm_bias_act_bwd's column budget) stay on torch modules and are
        single-process only."""
        ops = adj_low if isinstance(adj_low, FilterOperators) else None
        group = ops.group if (ops is not None and ops.sharded) else None
        if len(self.mlpX.lins) == 1 and self.mlpX.lins[0].out_features <= 256:      # (acm_bias_act_bwd's column budget)
            lin = self.mlpX.lins[0]
            return AF.residual_linear(x, lin.weight, lin.bias, relu=True, drop=drop, group=group, call=call, pipe=pipe)
        if group is not None:
            raise NotImplementedError("row-sharded acmgcnpp supports init_layers_X = 1 with nhid <= 256 (the torch fallback "
                                      "does not reduce its gradients over the ranks)")
        if isinstance(x, SparseFeatures):
            raise NotImplementedError("CSR features with init_layers_X > 1 or nhid > 256")
        f_in = self.mlpX.lins[0].in_features              # x may carry zero pad columns (dropout(..., pad_to=...))
        out = F.relu(self.mlpX(x if x.shape[1] == f_in else x[:, :f_in], input_tensor=True))
        if drop is not None and drop[0] > 0:
            out = AF.dropout(out, drop[0], drop[2], tag=drop[1], row_offset=drop[3])
        return out

    def _forward_fused_dropout(self, x, adj_low, adj_high, adj_low_unnormalized, call):
        """Training forward with every dropout drawn from ``dropout_state`` (tags: 0 input, 1 hidden, 2 the
        ACM-GCN++ residual branch); same structure as forward()."""
        p = self.dropout
        if self.dropout_state is None:
            dev = x.values.device if isinstance(x, SparseFeatures) else x.device
            self.dropout_state = AF.DropoutState(dev)
        st = self.dropout_state
        off = adj_low.row_offset if isinstance(adj_low, FilterOperators) else 0
        kw = {}
        if isinstance(x, SparseFeatures):
            x = x.with_values(AF.dropout(x.values.reshape(-1, 1), p, st, tag=0).reshape(-1))
        else:
            nfeat = x.shape[1]
            pad = AF.agg_pad_width(nfeat)
            ops = adj_low if isinstance(adj_low, FilterOperators) else None
            piped = (call.pipe is not None and call.pipe.primed and call.pipe.ops is ops and call.pipe.state is st
                     and call.pipe.x_rows.data_ptr() == x.data_ptr() and call.pipe.x_rows.shape == x.shape
                     and torch.is_grad_enabled())
            if piped:
                # dropout_t(x), drawn one step ahead (functional.InputPipeline); row-sharded: of every node, this rank's
                # rows a view of it
                x = call.pipe.local_table()
                if ops.sharded:
                    ops._pregathered = (x, call.pipe.table())
            elif ops is not None and ops.sharded and ops.uniform and ops.x_full is not None and nfeat <= 16:
                # (a first layer that GATHERS its input -- aggregate-first, F_in <= 16 -- needs every node's dropped row)
                # the mask is a function of the global position: drop the replicated full input locally instead of
                # all-gathering the dropped row blocks (equal blocks only: there the halo numbering is the global one).
                # A wider input is projected first, from the rank's OWN rows: those alone are dropped below (round 5: the
                # Penn94-shaped rank of an 8-rank plan spent 439 of its 640 us dropping the other ranks' 4 814-wide rows)
                xg = AF.dropout(ops.x_full, p, st, tag=0, pad_to=pad, row_offset=0)
                x = xg[off:off + x.shape[0]]
                ops._pregathered = (x, xg)
            elif (self.model_type in ("acmgcn", "acmgcnp", "acmsgc") and pad == nfeat and not x.requires_grad
                    and AF.in_drop_supported(x, ops, self.gcns[0]._config(), nfeat, self.gcns[0].out_features)
                    if ops is not None else False):
                # a wide dense input: the first layer's projection applies the input dropout while it stages X (forward
                # and backward); the dropped copy of X is never written
                kw = {"input_drop": (p, 0, st)}
            else:
                x = AF.dropout(x, p, st, tag=0, pad_to=pad, row_offset=off)
        if self.model_type == "acmsgc":
            return self.gcns[0](x, adj_low, adj_high, adj_low_unnormalized, rows_permuted=self._rows_permuted, call=call, **kw)
        xx = None
        lin = self.mlpX.lins[0] if (self.model_type == "acmgcnpp" and len(self.mlpX.lins) == 1) else None
        # the residual branch of a narrow dense input rides ONE launch behind the first layer (fea + xX: functional.
        # residual_add_linear, masks recomputed in its backward); every other case computes xX first, as the reference does
        add_fused = lin is not None and AF.residual_add_supported(x, lin.weight)
        if self.model_type == "acmgcnpp" and not add_fused:
            # (piped: x is the pipeline's table, which the first layer's forward below refills for the next step)
            xx = self._residual(x, adj_low, drop=(p, 2, st, off), call=call,
                                pipe=call.pipe if (not isinstance(x, SparseFeatures) and piped) else None)
        # the output layer's narrow projection may ride the hidden layer's epilogue (CallContext.next_proj / pre_proj); not
        # with the ACM-GCN++ residual, which changes the hidden activations in between
        call.next_proj = self.gcns[1] if self.model_type != "acmgcnpp" else None
        fea = self.gcns[0](x, adj_low, adj_high, adj_low_unnormalized, post_relu=True, post_drop=(p, 1, st), **kw,
                           rows_permuted=self._rows_permuted, call=call)
        call.next_proj = None
        if add_fused:
            ops_ = adj_low if isinstance(adj_low, FilterOperators) else None
            xr = x
            if call.pipe is not None and x.data_ptr() == call.pipe.local_table().data_ptr() and call.pipe.adopted:
                xr = call.pipe.saved[0]        # the table holds step t + 1's rows by now: this step's are in the saved copy
            fea = AF.residual_add_linear(fea, xr, lin.weight, lin.bias, relu=True, drop=(p, 2, st, off),
                                         group=ops_.group if (ops_ is not None and ops_.sharded) else None, call=call)
        elif self.model_type == "acmgcnpp":
            fea = fea + xx
        else:
            call.hidden_private = fea          # consumed by the output layer only: its gradient may stay implicit
        try:
            return self.gcns[1](fea, adj_low, adj_high, adj_low_unnormalized, rows_permuted=self._rows_permuted, call=call)
        finally:
            call.hidden_private = None

    def _forward_snowball(self, x, adj_low, adj_high, fused, call):
        """models.py:57-64: h_k = dropout(relu(layer_k([x | h_0 | ... | h_{k-1}]))), out = layer_last([x | h_0 | ...]).
        The ReLU + dropout after every hidden layer ride that layer's epilogue (post_relu / post_scale / post_drop);
        the concatenations are plain copies."""
        if isinstance(x, SparseFeatures):
            raise NotImplementedError("acmsnowball concatenates the input with the hidden blocks: dense features only")
        p, st = self.dropout, self.dropout_state
        off = adj_low.row_offset if isinstance(adj_low, FilterOperators) else 0
        if fused:
            x = AF.dropout(x, p, st, tag=0, row_offset=off)
        else:
            x = F.dropout(x, p, training=self.training)
        blocks = []
        for k in range(self.nlayers):
            inp = x if k == 0 else torch.cat([x] + blocks, 1)
            if fused:
                h = self.gcns[k](inp, adj_low, adj_high, None, post_relu=True, post_drop=(p, 1 + k, st), rows_permuted=self._rows_permuted, call=call)
            else:
                scale = None
                if self.training and p > 0:
                    scale = F.dropout(self._ones_like_hidden(x.shape[0], self.gcns[k].out_features, x.device), p, training=True)
                h = self.gcns[k](inp, adj_low, adj_high, None, post_relu=True, post_scale=scale, rows_permuted=self._rows_permuted, call=call)
            blocks.append(h)
        return self.gcns[-1](torch.cat([x] + blocks, 1), adj_low, adj_high, None, rows_permuted=self._rows_permuted, call=call)

    def auto_csr(self, x, ops):
        """Wide, mostly-zero features handed over dense -> their CSR twin (graph.SparseFeatures.auto, tuning key
        ``csr_features``), where this model has the CSR route: not acmsnowball (it concatenates the input with the hidden
        blocks), not an mlpX stack the residual kernel does not cover, not row-sharded operators (a rank's block keeps the
        dense halo exchange), and not while someone has replaced ``F.dropout`` (a mask-replay harness hands out masks of
        the dense shape: the input stays what the masks were recorded for)."""
        if not isinstance(x, torch.Tensor) or F.dropout is not _TORCH_DROPOUT or self.model_type == "acmsnowball":
            return x
        if ops is not None and ops.sharded:
            return x
        if self.model_type == "acmgcnpp" and not (len(self.mlpX.lins) == 1 and self.mlpX.lins[0].out_features <= 256):
            return x
        return SparseFeatures.auto(x)

    def forward(self, x, adj_low, adj_high=None, adj_low_unnormalized=None, rows_permuted=False, call=None):
        """Reference signature.  With relabelled operators (graph.relabel_by_degree) the rows are translated ONCE here
        -- x on the way in, the logits on the way out -- and every layer in between works in the relabelled numbering
        (``rows_permuted=True``: the caller, e.g. train.TrainStep, already did and wants the result there too).
        ``call``: a functional.CallContext for this forward (train.TrainStep passes its own); by default a fresh one that
        inherits what the calling thread's ``with functional.deferred_reductions() / fused_loss_tail()`` blocks set."""
        call = AF.CallContext.from_ambient() if call is None else call
        ops = adj_low if isinstance(adj_low, FilterOperators) else None
        if ops is None and isinstance(adj_low, torch.Tensor):
            from .graph import operators_for
            four = self.structure_info and self.model_type in ("acmgcnp", "acmgcnpp")
            ops = adj_low = operators_for(adj_low, adj_high, adj_low_unnormalized if four else None)
        self.__dict__["_rows_permuted"] = ops is not None and ops.perm is not None
        if not rows_permuted:                  # (a caller that already permuted -- train.TrainStep -- asked before it did)
            x = self.auto_csr(x, ops)
        if self._rows_permuted and not rows_permuted:
            x = x.permute_rows(ops.perm) if isinstance(x, SparseFeatures) else x.index_select(0, ops.perm)
            return self._forward(x, adj_low, adj_high, adj_low_unnormalized, call).index_select(0, ops.inv_perm)
        return self._forward(x, adj_low, adj_high, adj_low_unnormalized, call)

    def _forward(self, x, adj_low, adj_high, adj_low_unnormalized, call):
        fused = self.fused_dropout and self.training and self.dropout > 0
        if fused and self.dropout_state is None:
            dev = x.values.device if isinstance(x, SparseFeatures) else x.device
            self.dropout_state = AF.DropoutState(dev)
        if self.model_type == "acmsnowball":
            return self._forward_snowball(x, adj_low, adj_high, fused, call)
        if fused:
            return self._forward_fused_dropout(x, adj_low, adj_high, adj_low_unnormalized, call)
        drop = lambda t: F.dropout(t, self.dropout, training=self.training)  # noqa: E731
        if isinstance(x, SparseFeatures):
            # dropout of a sparse matrix = dropout of its stored values (zeros stay zero either way)
            x = x.with_values(drop(x.values))
        else:
            x = drop(x)
        if self.model_type == "acmsgc":
            return self.gcns[0](x, adj_low, adj_high, adj_low_unnormalized, rows_permuted=self._rows_permuted, call=call)
        if self.model_type == "acmgcnpp":
            xx = drop(self._residual(x, adj_low, call=call))
        # dropout(relu(fea1)) (models.py:70) rides the layer's epilogue: the keep-mask / (1 - p) tensor is what
        # F.dropout does to a tensor of ones, so a patched F.dropout (mask replay in tests) is honoured
        scale = None
        if self.training and self.dropout > 0:
            ones = self._ones_like_hidden(x.shape[0], self.gcns[0].out_features, x.device)
            scale = drop(ones)
        call.next_proj = self.gcns[1] if self.model_type != "acmgcnpp" else None     # see _forward_fused_dropout
        fea = self.gcns[0](x, adj_low, adj_high, adj_low_unnormalized, post_relu=True, post_scale=scale,
                           rows_permuted=self._rows_permuted, call=call)
        call.next_proj = None
        if self.model_type == "acmgcnpp":
            fea = fea + xx
        else:
            call.hidden_private = fea          # see _forward_fused_dropout
        try:
            return self.gcns[1](fea, adj_low, adj_high, adj_low_unnormalized, rows_permuted=self._rows_permuted, call=call)
        finally:
            call.hidden_private = None
