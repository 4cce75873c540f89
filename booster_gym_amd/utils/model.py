"""Actor-critic network.

Same architecture and `state_dict` key names as reference `utils/model.py:5-36` (so checkpoints and the exported
TorchScript actor interoperate): `critic.{0,2,4,6}`, `actor.{0,2,4,6}`, `logstd`; actor 47-256-128-128-12,
critic (47+14)-256-256-128-1, ELU, state-independent log-std initialised to -2.
GEMMs run through PyTorch-ROCm (hipBLASLt / rocBLAS, fp32 MFMA); the rollout-time inference + sampling is one
fused HIP launch (`sample_actions` -> bg_actor_sample).
"""
import ctypes

import torch

from .. import _lib

ACTOR_HIDDEN = (256, 128, 128)
CRITIC_HIDDEN = (256, 256, 128)


def _mlp(n_in, hidden, n_out):
    layers, prev = [], n_in
    for h in hidden:
        layers += [torch.nn.Linear(prev, h), torch.nn.ELU()]
        prev = h
    layers.append(torch.nn.Linear(prev, n_out))
    return torch.nn.Sequential(*layers)


def plan_wgrad_slices(shapes, rows, workgroups=256, share_rows=True):
    """(slices, tiles per workgroup) per layer for bg_mlp_weight_grad_group: `shapes` = [(C_out, C_in padded)].  One workgroup per (group of tw
    output tiles, slice); its 4 waves cover tw tiles x ks = 4 / tw sub-ranges of the slice's rows.  Every WAVE of the launch should do the same MFMA
    work: a wave's cost is rows / (slices ks) x tile width, so slices is proportional to width / ks; the total stays within `workgroups` (one
    512-register workgroup per CU).  share_rows: layers with 2 or 4 tiles put them in one workgroup (tw = tile count), so that the waves working on
    the same rows fetch them once per CU (less input traffic, more partial-tile traffic).  Largest-remainder rounding."""
    tiles = [(co // 128) * max(1, ci // 128) for co, ci in shapes]
    width = [0.5 if ci == 64 else 1.0 for _, ci in shapes]
    tw = [(t if t in (2, 4) else 1) if share_rows else 1 for t in tiles]
    ks = [4 // t for t in tw]
    units = sum(t * c for t, c in zip(tiles, width))  # total work in (128 x 128 tile) x rows
    scale = 4.0 * workgroups / units
    ideal = [scale * c / k for c, k in zip(width, ks)]
    groups = [t // w for t, w in zip(tiles, tw)]
    # every wave should get at least one run of 16 row pairs (fp32 kernel) / two 16-row blocks (split kernel: a hard limit there): slices x ks <=
    # rows / 32.  Below 32 x ks rows one slice remains and some of its waves get an empty run, which the fp32 kernel handles (they add zeros).
    cap = [max(1, rows // (32 * k)) for k in ks]
    s = [max(1, min(cp, int(x))) for x, cp in zip(ideal, cap)]
    order = sorted(range(len(shapes)), key=lambda k: ideal[k] - int(ideal[k]), reverse=True)
    for k in order:
        if s[k] < cap[k] and sum(g * v for g, v in zip(groups, s)) + groups[k] <= workgroups:
            s[k] += 1
    return s, tw


class GroupedWeightGrad:
    """All deferred weight gradients of several MLPTrainers in one launch pair (bg_mlp_weight_grad_group)."""

    def __init__(self, workgroups=None):
        self.workgroups = workgroups or MLPTrainer.WGRAD_WORKGROUPS
        # waves of a workgroup on the same rows, different tiles (plan_wgrad_slices): measured no faster alone (329.6 vs 329.2 us for the six layers) and
        # 0.3 ms slower per update in the loop (4x the partial-tile traffic for the 256 x 256 layer), so the pure split over rows stays the default
        self.share_rows = False
        self._key, self._arr, self._scratch = None, None, None
        self.timed_events = None

    def run(self, trainers, finish=True):
        """finish = False: the main kernel only (bg_mlp_weight_grad_group_partial); returns (descriptor array, count) for bg_update_tail, which sums the
        slices inside the mini-epoch's last launch -- or None where that form does not apply (split mode, no supported layer) and the finished
        gradients have been written as usual."""
        probs = [p for tr in trainers for p in tr.pending_wgrad_problems()]
        if not probs:
            return None
        key = (MLPTrainer.SPLIT, MLPTrainer.WGRAD_SPLIT) + tuple((g.data_ptr(), a.data_ptr(), dw.data_ptr(), g.shape[0], co, ci, cr) for g, a, dw, co, ci, cr in probs)
        if key != self._key:  # buffers are static: built once
            rows = probs[0][0].shape[0]
            # split mode (bg_mlp_weight_grad_group_split): the waves of a workgroup share their rows through LDS, all tiles of a layer in one workgroup
            # (MLPTrainer.WGRAD_SPLIT: the split launch behind the chained kernels, its finish inside bg_update_tail like the fp32 launch's)
            chained = all(tr._chain_split_bwd() for tr in trainers)
            self.split = (MLPTrainer.SPLIT or (MLPTrainer.WGRAD_SPLIT if chained else 0)) if rows % 32 == 0 and rows >= 128 and all((co, ci) in ((256, 256), (128, 256), (128, 128), (256, 64)) for _, _, _, co, ci, _ in probs) else 0
            slices, tw = plan_wgrad_slices([(co, ci) for _, _, _, co, ci, _ in probs], rows, self.workgroups, share_rows=self.share_rows or bool(self.split))
            self._scratch = [torch.empty(sl * co * ci, dtype=torch.float32, device=probs[0][0].device) for sl, (_, _, _, co, ci, _) in zip(slices, probs)]
            arr = (_lib.WgradProblem * len(probs))()
            for k, ((g, a, dw, co, ci, cr), sl) in enumerate(zip(probs, slices)):
                arr[k].G, arr[k].A, arr[k].dW, arr[k].scratch = g.data_ptr(), a.data_ptr(), dw.data_ptr(), self._scratch[k].data_ptr()
                arr[k].M, arr[k].C_out, arr[k].C_in, arr[k].C_in_real, arr[k].slices, arr[k].tiles_per_workgroup = g.shape[0], co, ci, cr, sl, tw[k]
            self._key, self._arr, self.slices, self.tw = key, arr, slices, tw
        ev = self.timed_events
        if ev is not None:  # bench.py: HIP events on the launch stream around the launch pair
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        partial = not finish and (not self.split or not MLPTrainer.SPLIT)
        if self.split and partial:
            _lib.check(_lib.load().bg_mlp_weight_grad_group_split_partial(self._arr, len(probs), self.split, _lib.current_stream_ptr()), "bg_mlp_weight_grad_group_split_partial")
        elif self.split:
            _lib.check(_lib.load().bg_mlp_weight_grad_group_split(self._arr, len(probs), self.split, _lib.current_stream_ptr()), "bg_mlp_weight_grad_group_split")
        elif partial:
            _lib.check(_lib.load().bg_mlp_weight_grad_group_partial(self._arr, len(probs), _lib.current_stream_ptr()), "bg_mlp_weight_grad_group_partial")
        else:
            _lib.check(_lib.load().bg_mlp_weight_grad_group(self._arr, len(probs), _lib.current_stream_ptr()), "bg_mlp_weight_grad_group")
        if ev is not None:
            e1.record()
            # algorithmic flops: the REAL input columns (47 / 61 of the zero-padded 64 of the first layers)
            ev.append((e0, e1, sum(2.0 * g.shape[0] * co * cr for g, _, _, co, _, cr in probs), [(g.shape[0], co, cr) for g, _, _, co, _, cr in probs]))
        return (self._arr, len(probs)) if partial else None


class MLPTrainer:
    """Hand-scheduled forward / backward of one of the two ELU MLPs for the full-batch PPO update (replaces autograd for
    reference utils/runner.py:132,147,163).  Same arithmetic as torch's Linear/ELU autograd, different schedule:
      * forward: addmm (bias in the GEMM epilogue) + in-place ELU, activations kept for the backward;
      * backward per layer: one fused pass `g <- g * elu'(a)` + bias gradient (bg_elu_backward_colsum), the weight gradient as a
        split-K batched GEMM (`bmm` over S slices of the batch, then a sum) -- hipBLASLt's single GEMM with K = 98,304 runs at
        8-45 TF/s, the split form at 60-110 TF/s on MI355X -- and dX = g @ W;
      * gradients are WRITTEN into the parameters' `.grad` views of the flat Adam buffer (no AccumulateGrad adds, no zero_grad).
    """

    # Hidden layers with K in {64, 128, 256} and N % 128 == 0 run on the hand-written fused fp32-MFMA layer (bg_mlp.hip: bias + ELU in the GEMM
    # epilogue).  Measured on MI355X at M = 98,304 (tools/mlp_probe.py): 131.8 vs 151.3 us (256x256), 63.9 vs 76.3 us (256x128), 37.5 vs 55.8 us
    # (128x128) against hipBLASLt addmm + elu_.  Other shapes, or FUSED = False (a class attribute, for the tests that compare the two forms): the
    # library GEMM + elementwise ELU.
    FUSED = True

    # Opt-in (BG_GEMM_SPLIT=9 or 6): the same layers on the bf16 matrix pipe, every fp32 operand split exactly into three bf16 numbers and all 9
    # (or the 6 largest) cross products accumulated in fp32 (bg_mlp_split.hip).  0 = the fp32 MFMA kernels.
    SPLIT = int(__import__("os").environ.get("BG_GEMM_SPLIT", "0"))
    # The grouped weight-gradient launch on the bf16 matrix pipe as well (bg_wgrad_split.hip: exact 3-way splits, all 9 products, the sub-ranges of the
    # batch alternating the sign of the accumulation; its error against float64 is 0.84-0.89 of the fp32-MFMA launch's and it is 0.4 ms per iteration
    # faster in the loop: profiles/r06_wgrad_split_*).  BG_WGRAD_SPLIT=0: the fp32-MFMA launch (bg_wgrad.hip).  Applies behind the chained split kernels
    # only (GroupedWeightGrad.run); shapes outside the split kernel's four take the fp32 launch.
    WGRAD_SPLIT = int(__import__("os").environ.get("BG_WGRAD_SPLIT", "9"))

    @classmethod
    def _fusable(cls, k_in, n_out):
        return cls.FUSED and k_in in (64, 128, 256) and n_out % 128 == 0

    # The forward pass of the three hidden layers as one launch (bg_mlp_chain_forward) where the widths are the reference's (CHAIN = False: one launch
    # per layer)
    CHAIN = True

    # ... and that launch on the bf16 matrix pipe with fp32 semantics (bg_mlp_chain_split.hip: every fp32 operand the exact sum of three bf16 numbers, all
    # 9 cross products accumulated in fp32 -- 9 x 32 cycles per 32 x 32 x 16 block against 8 x 64 on the fp32 pipe, error against float64 at or below the
    # fp32-MFMA chain's).  The default; BG_CHAIN_SPLIT=0 (or CHAIN_SPLIT = False) runs the fp32-MFMA chain (bg_mlp_chain.hip).
    CHAIN_SPLIT = __import__("os").environ.get("BG_CHAIN_SPLIT", "1") == "1"
    # ... and the backward-data pass of the hidden layers as one launch of the same arithmetic (bg_mlp_chain_split_bwd.hip); BG_CHAIN_SPLIT_BWD=0: one
    # fp32-MFMA launch per layer (bg_mlp_layer_backward)
    CHAIN_SPLIT_BWD = __import__("os").environ.get("BG_CHAIN_SPLIT_BWD", "1") == "1"
    # Odd 128-row slabs of the chained split kernels accumulate the NEGATED sums (planes of -W beside the planes of W) and put the sign back where a tile
    # is finished: the bf16 MFMA's accumulator does not round to nearest, every accumulated element carries a small bias of one sign, and what is summed
    # over the rows downstream (bias gradients, weight gradients) would collect it; alternating makes it cancel.  BG_CHAIN_ALTERNATE=0: off.
    CHAIN_ALTERNATE = __import__("os").environ.get("BG_CHAIN_ALTERNATE", "1") == "1"

    def _chain_split(self):
        return self.CHAIN_SPLIT and self._chainable()

    def _fresh_planes(self):
        """The bf16 planes of the three hidden layers' weights (what the chained split kernel reads): created on first use; rewritten from the
        parameters unless the optimiser launch keeps them current (mirror_fresh)."""
        ls, lib, stream = self.layers, _lib.load(), _lib.current_stream_ptr()
        new = self.cplanes[0] is None
        for i in range(3):
            n_out, k_in = ls[i].weight.shape
            kp = self._kin if i == 0 else k_in
            if new:  # (the planes of W, then the planes of -W)
                self.cplanes[i] = torch.zeros(2 * n_out * kp * 3, dtype=torch.int16, device=ls[i].weight.device)
            if new or not self.mirror_fresh:
                _lib.check(lib.bg_mlp_split_weights_pm(n_out, kp, _lib.ptr(ls[i].weight), k_in, n_out, k_in, 0, _lib.ptr(self.cplanes[i]), stream), "bg_mlp_split_weights_pm")
        return self.cplanes

    def _chain_split_bwd(self):
        """Does backward_hidden run as ONE launch on the bf16 matrix pipe (bg_mlp_chain_split_bwd.hip)?  Same widths as the chained forward."""
        return self.CHAIN_SPLIT_BWD and self._chain_split()

    def _fresh_planes_t(self):
        """The bf16 planes of the transposed weights of hidden layers 2 and 1 (what the chained backward kernel reads), as _fresh_planes."""
        ls, lib, stream = self.layers, _lib.load(), _lib.current_stream_ptr()
        new = self.cplanes_t[1] is None
        for i in (1, 2):
            c_out, c_in = ls[i].weight.shape
            if new:
                self.cplanes_t[i] = torch.zeros(2 * c_in * c_out * 3, dtype=torch.int16, device=ls[i].weight.device)
            if new or not self.mirror_fresh:
                _lib.check(lib.bg_mlp_split_weights_pm(c_in, c_out, _lib.ptr(ls[i].weight), c_in, c_out, c_in, 1, _lib.ptr(self.cplanes_t[i]), stream), "bg_mlp_split_weights_pm")
        return self.cplanes_t

    def chain_backward_descriptor(self, g=None):
        """bg_mlp_chain_split_bwd of this network's backward-data pass from g = dL/dz of the last hidden layer (default: hidden_grad)."""
        ls, B, p = self.layers, self._B, _lib.ptr
        g = self.hidden_grad if g is None else g
        PT = self._fresh_planes_t()
        n1, n2, n3 = ls[0].weight.shape[0], ls[1].weight.shape[0], ls[2].weight.shape[0]
        slabs = (B + 127) // 128
        wg = self.chain_bwd_workgroups if 0 < self.chain_bwd_workgroups < slabs else slabs
        if self.chain_colsum is None or self.chain_colsum.numel() < slabs * 4 * (n1 + n2):  # one record of column sums per (slab, wave)
            self.chain_colsum = torch.empty(slabs * 4 * (n1 + n2), dtype=torch.float32, device=g.device)
        self._pending_wgrad = [(2, g), (1, self.gin[2]), (0, self.gin[1])]
        return _lib.MlpChainSplitBwd(B, n1, n2, n3, int(self.chain_bwd_workgroups), int(self.CHAIN_ALTERNATE), p(g), p(PT[2]), p(PT[1]), p(self.acts[1]), p(self.acts[0]), p(self.gin[2]),
                                     p(self.gin[1]), p(self.chain_colsum), p(ls[1].bias.grad), p(ls[0].bias.grad))

    def _chainable(self):
        ls = self.layers
        return (self.CHAIN and self.FUSED and not self.SPLIT and len(ls) == 4 and self._kin == 64 and self.w0pad is not None
                and tuple(l.weight.shape[0] for l in ls[:3]) in ((256, 128, 128), (256, 256, 128))
                and ls[1].weight.shape[1] == ls[0].weight.shape[0] and ls[2].weight.shape[1] == ls[1].weight.shape[0])

    def chainable_for(self, x, train_rows=None):
        """Will forward_hidden(x, train_rows) run the chained kernel (and with it a `value_head`)?"""
        B = x.shape[0] if train_rows is None else train_rows
        if self._B != B or self._rows != x.shape[0] or self._kin != x.shape[1]:
            self._alloc(x.shape[0], B, x.device, x.shape[1])
        return self._chainable()

    def _chain_descriptor(self):
        """bg_mlp_chain of this network's hidden layers on the input of the forward pass in progress (self.x)."""
        ls = self.layers
        p = _lib.ptr
        vw, vb, vo = self.value_head if self.value_head is not None else (None, None, None)
        if vo is not None and (vo.numel() < self.x.shape[0] or vw.numel() != ls[2].weight.shape[0]):
            raise ValueError("value_head: weight [width of the last hidden layer], bias [1], output [rows]")
        if self._chain_split():
            P = self._fresh_planes()
            return _lib.MlpChainSplit(self.x.shape[0], self._kin, ls[0].weight.shape[0], ls[1].weight.shape[0], ls[2].weight.shape[0], int(self.chain_workgroups),
                                      int(self.CHAIN_ALTERNATE), 0, p(self.x), p(P[0]), p(P[1]), p(P[2]), p(ls[0].bias), p(ls[1].bias), p(ls[2].bias), p(self.acts[0]), p(self.acts[1]),
                                      p(self.acts[2]), p(vw), p(vb), p(vo))
        if not self.mirror_fresh:
            self.w0pad[:, : ls[0].weight.shape[1]].copy_(ls[0].weight)
        return _lib.MlpChain(self.x.shape[0], self._kin, ls[0].weight.shape[0], ls[1].weight.shape[0], ls[2].weight.shape[0], int(self.chain_workgroups), p(self.x),
                             p(self.w0pad),
                             p(ls[0].bias), p(ls[1].weight), p(ls[1].bias), p(ls[2].weight), p(ls[2].bias), p(self.acts[0]), p(self.acts[1]), p(self.acts[2]),
                             p(vw), p(vb), p(vo))

    def prepare(self, x, train_rows=None):
        """Workspaces for a forward pass on x (first `train_rows` rows = the batch the backward pass differentiates) without running it."""
        B = x.shape[0] if train_rows is None else train_rows
        if self._B != B or self._rows != x.shape[0] or self._kin != x.shape[1]:
            self._alloc(x.shape[0], B, x.device, x.shape[1])
        self.x = x

    def refresh_mirrors(self):
        """Rewrite the copies of the weights that the layer kernels read (zero-padded first layer, transposed hidden layers) from the parameters, on
        the current stream.  The optimiser launch keeps them current; this covers weights changed by other means (checkpoint, broadcast, a test)."""
        ls = self.layers
        if self.w0pad is not None:
            self.w0pad[:, : ls[0].weight.shape[1]].copy_(ls[0].weight)
        for i, w in enumerate(self.wt):
            if w is not None:
                w.copy_(ls[i].weight.t())
        if self.cplanes[0] is not None or self.cplanes_t[1] is not None:
            self.mirror_fresh = False
            if self.cplanes[0] is not None:
                self._fresh_planes()
            if self.cplanes_t[1] is not None:
                self._fresh_planes_t()
        self.mirror_fresh = True

    def chain_rows_descriptor(self, row0, nrows):
        """bg_mlp_chain of rows [row0, row0 + nrows) of the pass prepared by `prepare` (whole 128-row slabs: the kernel stores every slab in full):
        the same launch the full-batch forward makes, restricted to these slabs -- bit-identical outputs in the same places of the activation buffers
        (and of the value head's output).  Used by the rollout, which evaluates each step's rows as soon as the simulator has produced them."""
        if row0 % 128 or nrows % 128 or row0 + nrows > self.x.shape[0]:
            raise ValueError("chain_rows_descriptor: row0 and nrows must be multiples of 128 inside the prepared batch")
        d = self._chain_descriptor()
        d.M = nrows
        d.workgroups = 0  # (a few slabs during the rollout: one workgroup each)
        d.X = d.X + 4 * row0 * self._kin
        ls = self.layers
        d.Y1, d.Y2, d.Y3 = (y + 4 * row0 * l.weight.shape[0] for y, l in zip((d.Y1, d.Y2, d.Y3), ls[:3]))
        if d.v_out:
            d.v_out = d.v_out + 4 * row0
        return d

    @staticmethod
    def forward_rows_group(jobs):
        """jobs = [(trainer, row0, nrows), ...]: the chained forward of those rows of every trainer's prepared pass in ONE launch (at most 4)."""
        descs = [tr.chain_rows_descriptor(r0, nr) for tr, r0, nr in jobs]
        MLPTrainer.launch_chain(descs)

    @staticmethod
    def forward_hidden_group(jobs):
        """jobs = [(trainer, x, train_rows), ...] (at most 4, all on the chained split kernel): `forward_hidden` of every job in ONE launch -- the
        networks share the chip by their `chain_workgroups` inside one grid instead of as launches on several streams.  Same kernel code per network,
        same slabs, same order of the sums: bit-identical to the separate launches.  Returns the last hidden activations of every job."""
        descs = []
        for tr, x, train_rows in jobs:
            tr.prepare(x, train_rows)
            if not tr._chain_split():
                raise ValueError("forward_hidden_group: every network must run the chained split kernel")
            descs.append(tr._chain_descriptor())
        timed = any(tr.timed_layer is not None for tr, _, _ in jobs)
        if timed:  # bench.py: HIP events on the launch stream around this one kernel (the same pair is noted for every network of the launch)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        MLPTrainer.launch_chain(descs)
        if timed:
            e1.record()
            for tr, x, _ in jobs:
                tr.timed_events.append((e0, e1, x.shape[0], tr._kin, tuple(l.weight.shape[0] for l in tr.layers[:3]), "chain_split"))
        return [tr.acts[2] for tr, _, _ in jobs]

    @staticmethod
    def backward_hidden_group(trainers, finishes):
        """`backward_hidden(finishes=finishes)` of every trainer (at most 4, all on the chained split backward kernel) in ONE launch; appends one
        reduction descriptor per trainer, in the order given."""
        if not all(tr._chain_split_bwd() for tr in trainers):
            raise ValueError("backward_hidden_group: every network must run the chained split backward kernel")
        timed = any(tr.timed_layer is not None for tr in trainers)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        ds = [tr.chain_backward_descriptor() for tr in trainers]
        arr = (_lib.MlpChainSplitBwd * len(ds))(*ds)
        fins = (_lib.ReduceProblem * len(ds))()
        _lib.check(_lib.load().bg_mlp_chain_backward_split(ctypes.addressof(arr), len(ds), fins, _lib.current_stream_ptr()), "bg_mlp_chain_backward_split")
        finishes.extend(fins[k] for k in range(len(ds)))
        if timed:
            e1.record()
            for tr in trainers:
                fl = 2.0 * tr._B * sum(l.weight.shape[0] * l.weight.shape[1] for l in tr.layers[1:-1])
                tr.timed_events.append((e0, e1, tr._B, fl, None, "backward"))

    @staticmethod
    def launch_chain(descs):
        """One launch for a list of chain descriptors of one kind (all bg_mlp_chain or all bg_mlp_chain_split); a mixed list runs as two launches."""
        lib, st = _lib.load(), _lib.current_stream_ptr()
        for kind, fn, name in ((_lib.MlpChainSplit, lib.bg_mlp_chain_forward_split, "bg_mlp_chain_forward_split"),
                               (_lib.MlpChain, lib.bg_mlp_chain_forward_group, "bg_mlp_chain_forward_group")):
            sel = [d for d in descs if isinstance(d, kind)]
            if sel:
                arr = (kind * len(sel))(*sel)
                _lib.check(fn(ctypes.addressof(arr), len(sel), st), name)

    def __init__(self, seq, max_split=32):
        self.layers = [m for m in seq if isinstance(m, torch.nn.Linear)]
        self.max_split = max_split
        self.x = None
        self._B = self._rows = self._kin = None
        self.timed_layer, self.timed_events = None, []
        # workgroups of the full-batch chained forward launch (0: one per slab); the runner sets it when two networks' launches share the chip
        self.chain_workgroups = 0
        self.chain_bwd_workgroups = 0  # ... of the chained backward launch
        # (weight [N3], bias [1], out [rows]): a scalar output layer evaluated by the chained forward kernel itself (the critic's values); None: not
        self.value_head = None

    def _split(self, B):
        s = self.max_split
        while s > 1 and B % s:
            s -= 1
        return s

    def _alloc(self, rows, B, dev, k_in):
        """Static workspaces (no allocator traffic inside the update loop; safe to use from a side stream).  `rows` >= B: extra inference-only
        rows may ride along in the forward pass (the critic evaluates the T+1'th observation in the same GEMMs).  `k_in` > in_features of
        the first layer means the caller zero-padded the input columns (61 -> 64, 47 -> 64) so that the first layer, too, runs on the fused kernel."""
        self._rows, self._B, self._S, self._kin = rows, B, self._split(B), k_in
        # (rows rounded up to whole 128-row slabs: the chained forward kernel stores every slab in full)
        rows_pad = (rows + 127) // 128 * 128
        self.acts = [torch.empty(rows_pad, l.weight.shape[0], dtype=torch.float32, device=dev)[:rows] for l in self.layers]
        B_pad = (B + 127) // 128 * 128  # (whole 128-row slabs: the chained backward kernel stores every slab in full)
        self.gin = [None] + [torch.empty(B_pad, l.weight.shape[1], dtype=torch.float32, device=dev)[:B] for l in self.layers[1:]]
        self.cs = [torch.empty(((B + 127) // 128) * l.weight.shape[0], dtype=torch.float32, device=dev) for l in self.layers]
        # weight gradients: hand-written split-over-the-batch MFMA kernel (bg_mlp_weight_grad) where the shape allows, scratch = slices x dW
        self.wg_slices = [self._wgrad_slices(B, l.weight.shape[0], k_in if i == 0 else l.weight.shape[1]) for i, l in enumerate(self.layers)]
        self.dw = [torch.empty(self.wg_slices[i] or self._S, l.weight.shape[0], k_in if i == 0 else l.weight.shape[1], dtype=torch.float32, device=dev)
                   for i, l in enumerate(self.layers)]
        self.wt = [None] * len(self.layers)  # transposed weights for the fused backward kernel
        # True while w0pad and wt ARE the current weights: the optimiser launch keeps them current (mirror_descriptors); False makes forward /
        # backward copy them first.  The owner of the optimiser sets it (utils/runner.py) and clears it wherever weights change by other means.
        self.mirror_fresh = False
        self.cplanes = [None] * 3  # CHAIN_SPLIT: bf16 planes of the three hidden layers' weights for the chained forward kernel
        self.cplanes_t = [None] * 3  # ... and of the transposed weights of layers 1 and 2 for the chained backward kernel
        self.chain_colsum = None     # ... whose waves leave one record of column sums per slab here
        self.planes = [None] * len(self.layers)  # SPLIT: bf16 planes of the weights (forward) ...
        self.planes_t = [None] * len(self.layers)  # ... and of the transposed weights (backward)
        l0 = self.layers[0]
        self.w0pad = torch.zeros(l0.weight.shape[0], k_in, dtype=torch.float32, device=dev) if k_in != l0.weight.shape[1] else None
        self.dw0sum = torch.empty(l0.weight.shape[0], k_in, dtype=torch.float32, device=dev) if self.w0pad is not None else None

    # Weight gradients (the dW part of loss.backward(), runner.py:163): hand-written fp32-MFMA kernel, ALL hidden layers of both networks in one
    # launch pair after both backward chains (bg_mlp_weight_grad_group, GroupedWeightGrad).  Measured on MI355X, round 2
    # (tools/archive/ab_defer.sh, update phase per iteration): library split-K bmm + sum inside the chains 24.13 ms; the hand-written kernel one layer at a
    # time inside the chains 26.46 ms (its 512-register, 128 KB-LDS workgroups cannot share a CU with the other stream's kernels); the library
    # path deferred 24.39 ms; the grouped hand-written launch 23.01 ms = 3.77 M env-steps/s against 3.61 M.  FUSED_WGRAD = False selects the library path.
    FUSED_WGRAD = True
    WGRAD_WORKGROUPS = 256  # one 4-wave workgroup per CU

    @classmethod
    def _wgrad_slices(cls, B, c_out, c_in):
        """Number of batch slices for bg_mlp_weight_grad (0 = shape not supported: library GEMM).  One workgroup per (128 x 128 output tile,
        slice); slices a multiple of 8 (same-slice tiles share an XCD); runs of 16 row pairs per wave keep the MFMA loop free of idle trips."""
        if not (cls.FUSED_WGRAD and cls.FUSED) or c_out % 128 or (c_in != 64 and c_in % 128) or B % 2 or B < 64:
            return 0
        ntiles = (c_out // 128) * max(1, c_in // 128)
        s = max(8, cls.WGRAD_WORKGROUPS // ntiles // 8 * 8)
        while s > 8 and s * 4 > (B // 2) // 16:  # 4 waves per slice, each at least one run of 16 row pairs where the batch allows it
            s -= 8
        return s

    def forward_hidden(self, x, train_rows=None):
        """All layers but the output layer: returns the activations of the last hidden (ELU) layer [rows, width].  The output layer then runs
        fused with the loss (bg_actor_head / bg_critic_head_*), and `backward_hidden` takes over from the gradient those kernels produce."""
        return self.forward(x, train_rows, _stop_before_output=True)

    def forward(self, x, train_rows=None, _stop_before_output=False):
        """x [rows, in (possibly zero-padded)].  The first `train_rows` rows (default: all) are the batch the backward pass differentiates."""
        B = x.shape[0] if train_rows is None else train_rows
        if self._B != B or self._rows != x.shape[0] or self._kin != x.shape[1]:
            self._alloc(x.shape[0], B, x.device, x.shape[1])
        self.x, h = x, x
        last = len(self.layers) - 1
        lib, stream = _lib.load(), _lib.current_stream_ptr()
        first = 0
        if self._chainable():
            # the three hidden layers in ONE launch, activations handed on in registers (bg_mlp_chain.hip); bit-identical to the per-layer launches
            timed = self.timed_layer is not None
            if timed:  # bench.py: HIP events on the launch stream around this one kernel
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            d = self._chain_descriptor()
            self.launch_chain([d])
            if timed:
                e1.record()
                self.timed_events.append((e0, e1, x.shape[0], self._kin, tuple(l.weight.shape[0] for l in self.layers[:3]), "chain_split" if self._chain_split() else "chain"))
            first, h = 3, self.acts[2]
        for i, l in enumerate(self.layers):
            if i < first:
                continue
            if i == last and _stop_before_output:
                break
            n_out, k_in = l.weight.shape
            w = l.weight
            if self.SPLIT and i < last and self._fusable(self._kin if i == 0 else k_in, n_out):
                kp = self._kin if i == 0 else k_in
                if self.planes[i] is None:
                    self.planes[i] = torch.empty(n_out * kp * 3, dtype=torch.int16, device=h.device)
                _lib.check(lib.bg_mlp_split_weights(n_out, kp, _lib.ptr(l.weight), k_in, n_out, k_in, 0, _lib.ptr(self.planes[i]), stream), "bg_mlp_split_weights")
                _lib.check(lib.bg_mlp_layer_forward_split(h.shape[0], kp, n_out, _lib.ptr(h), _lib.ptr(self.planes[i]), _lib.ptr(l.bias), _lib.ptr(self.acts[i]),
                                                          1, self.SPLIT, stream), "bg_mlp_layer_forward_split")
                h = self.acts[i]
                continue
            if i == 0 and self.w0pad is not None:
                if not self.mirror_fresh:
                    self.w0pad[:, :k_in].copy_(l.weight)  # weights change every optimiser step; 16k floats
                w, k_in = self.w0pad, self._kin
            if i < last and self._fusable(k_in, n_out):
                # hand-written fp32-MFMA layer with bias + ELU in the epilogue (bg_mlp.hip)
                timed = self.timed_layer is not None and (i == self.timed_layer or (isinstance(self.timed_layer, (tuple, list, set)) and i in self.timed_layer))
                if timed:  # bench.py: HIP events on the launch stream around this one kernel
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                _lib.check(lib.bg_mlp_layer_forward(h.shape[0], k_in, n_out, _lib.ptr(h), _lib.ptr(w), _lib.ptr(l.bias), _lib.ptr(self.acts[i]), 1,
                                                    stream), "bg_mlp_layer_forward")
                if timed:
                    e1.record()
                    self.timed_events.append((e0, e1, h.shape[0], k_in, n_out, i))
                h = self.acts[i]
                continue
            torch.addmm(l.bias, h, w.t(), out=self.acts[i])
            h = self.acts[i]
            if i < last:
                torch.nn.functional.elu_(h)
        return h

    def backward(self, grad_out):
        """grad_out [B, out] is consumed (modified in place).  Fills weight.grad / bias.grad of every layer.

        g always holds dL/dz of layer i (z = pre-activation).  For the linear output layer that is grad_out itself; going down, the fused
        kernel bg_mlp_layer_backward produces dL/dz of layer i-1 = (g W_i) * elu'(a_{i-1}) together with layer i-1's bias gradient in one
        pass; the skinny output layers (12 / 1 columns) use torch.mm + the fused ELU-backward/column-sum kernel instead."""
        last = len(self.layers) - 1
        torch.sum(grad_out, dim=0, out=self.layers[last].bias.grad)  # linear output layer: plain column sum
        self._backward_from(last, grad_out)

    @property
    def hidden_grad(self):
        """[B, width] buffer the fused head kernels write dL/dz of the last hidden layer into (input of `backward_hidden`)."""
        return self.gin[len(self.layers) - 1]

    def backward_hidden(self, g=None, finishes=None):
        """Backward from the last hidden layer down.  g = dL/dz of that layer (default: `hidden_grad`); its bias gradient and the output
        layer's weight / bias gradients have already been written by the fused head kernel.  finishes (a list): the fused backward layers run
        without their column-sum finish and append its descriptor (_lib.ReduceProblem) instead; the caller runs them later with
        utils.reduce_group (the bias gradients are not needed before the optimiser step)."""
        timed = self.timed_layer is not None
        if timed:  # bench.py: HIP events on the launch stream around this network's backward-data chain
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if self._chain_split_bwd():
            d = self.chain_backward_descriptor(g)
            fin = _lib.ReduceProblem()
            _lib.check(_lib.load().bg_mlp_chain_backward_split(ctypes.addressof(d), 1, fin, _lib.current_stream_ptr()), "bg_mlp_chain_backward_split")
            if finishes is not None:
                finishes.append(fin)
            else:
                from .utils import reduce_group
                reduce_group([fin])
        else:
            self._backward_from(len(self.layers) - 2, self.hidden_grad if g is None else g, finishes)
        if timed:
            e1.record()
            # dX = G W of every hidden layer but the first (whose input gradient nobody needs): 2 B C_out C_in each
            fl = 2.0 * self._B * sum(l.weight.shape[0] * l.weight.shape[1] for l in self.layers[1:-1])
            self.timed_events.append((e0, e1, self._B, fl, None, "backward"))

    # The backward chain computes only dL/dz; the weight gradients of all layers run afterwards, when both networks' chains are done and nothing
    # else competes for the GPU: the grouped launch (GroupedWeightGrad.run); shapes outside its range run as library GEMMs there.

    def pending_wgrad_problems(self):
        """(G, A, dW, C_out, C_in (padded), C_in_real) of every deferred weight gradient, and clears the list (for the grouped launch)."""
        out = []
        for i, g in self._pending_wgrad:
            if not self.wg_slices[i]:  # shape outside the kernel's range: the library path, now
                self._weight_grad(i, g)
                continue
            l = self.layers[i]
            a_in = (self.acts[i - 1] if i > 0 else self.x)[: self._B]
            out.append((g, a_in, l.weight.grad, l.weight.shape[0], a_in.shape[1], l.weight.shape[1]))
        self._pending_wgrad = []
        return out

    def _weight_grad(self, i, g):
        lib, stream = _lib.load(), _lib.current_stream_ptr()
        B, S = self._B, self._S
        l = self.layers[i]
        a_in = (self.acts[i - 1] if i > 0 else self.x)[:B]
        C_out, C_in = l.weight.shape
        if self.wg_slices[i]:
            # dW = G^T A in one hand-written launch pair, written straight into the flat gradient buffer (padded input columns dropped)
            _lib.check(lib.bg_mlp_weight_grad(B, C_out, a_in.shape[1], C_in, _lib.ptr(g), _lib.ptr(a_in), _lib.ptr(l.weight.grad), _lib.ptr(self.dw[i]),
                                              self.wg_slices[i], stream), "bg_mlp_weight_grad")
        elif i == 0 and self.w0pad is not None:  # padded input columns: their gradient columns are dropped
            torch.bmm(g.view(S, B // S, C_out).transpose(1, 2), a_in.view(S, B // S, self._kin), out=self.dw[0])
            torch.sum(self.dw[0], dim=0, out=self.dw0sum)
            l.weight.grad.copy_(self.dw0sum[:, :C_in])
        else:
            torch.bmm(g.view(S, B // S, C_out).transpose(1, 2), a_in.view(S, B // S, C_in), out=self.dw[i])
            torch.sum(self.dw[i], dim=0, out=l.weight.grad)

    def mirror_descriptors(self, flat):
        """bg_param_mirror entries for the copies of this network's weights that the layer kernels read (the bf16 planes of the chained split forward
        or the zero-padded first layer of the fp32 chain; the transposed hidden layers of the backward kernels): handed to bg_optimizer_step, which then writes them together with the parameters.  `flat`: the optimiser's flat
        parameter buffer (the weights are views of it).  Only buffers that exist are listed (they are created by the first forward / backward)."""
        out = []
        if self.SPLIT or not self.FUSED:
            return out
        split = self._chain_split() and self.cplanes[0] is not None
        for i, l in enumerate(self.layers):
            off = (l.weight.data_ptr() - flat.data_ptr()) // 4
            rows, cols = l.weight.shape
            if split and i < 3:  # the chained split kernel's planes (the padded input columns of the first layer stay zero)
                kp = self._kin if i == 0 else cols
                out.append(_lib.ParamMirror(off, rows, cols, 2, kp, rows * kp * 3, _lib.ptr(self.cplanes[i])))  # (pad: the planes of -W behind)
            if i == 0 and self.w0pad is not None and not split:
                out.append(_lib.ParamMirror(off, rows, cols, 0, self.w0pad.shape[1], 0, _lib.ptr(self.w0pad)))
            if i in (1, 2) and self.cplanes_t[i] is not None:  # the chained backward kernel's planes of W^T
                out.append(_lib.ParamMirror(off, rows, cols, 3, rows, rows * cols * 3, _lib.ptr(self.cplanes_t[i])))
            if self.wt[i] is not None:
                out.append(_lib.ParamMirror(off, rows, cols, 1, rows, 0, _lib.ptr(self.wt[i])))
        return out

    def _backward_from(self, start, g, finishes=None):
        lib = _lib.load()
        B = self._B
        stream = _lib.current_stream_ptr()
        self._pending_wgrad = []
        for i in range(start, -1, -1):
            l = self.layers[i]
            a_in = (self.acts[i - 1] if i > 0 else self.x)[:B]
            C_out, C_in = l.weight.shape
            self._pending_wgrad.append((i, g))
            if i > 0:
                below = self.layers[i - 1]
                if self.SPLIT and self.FUSED and C_out in (128, 256) and C_in % 128 == 0:
                    if self.planes_t[i] is None:
                        self.planes_t[i] = torch.empty(C_in * C_out * 3, dtype=torch.int16, device=g.device)
                    _lib.check(lib.bg_mlp_split_weights(C_in, C_out, _lib.ptr(l.weight), C_in, C_out, C_in, 1, _lib.ptr(self.planes_t[i]), stream),
                               "bg_mlp_split_weights")
                    _lib.check(lib.bg_mlp_layer_backward_split(B, C_out, C_in, _lib.ptr(g), _lib.ptr(self.planes_t[i]), _lib.ptr(a_in), _lib.ptr(self.gin[i]),
                                                               _lib.ptr(below.bias.grad), _lib.ptr(self.cs[i - 1]), self.SPLIT, stream),
                               "bg_mlp_layer_backward_split")
                elif self.FUSED and C_out in (128, 256) and C_in % 128 == 0:
                    if self.wt[i] is None:
                        self.wt[i] = torch.empty(C_in, C_out, dtype=torch.float32, device=g.device)
                        self.wt[i].copy_(l.weight.t())
                    elif not self.mirror_fresh:
                        self.wt[i].copy_(l.weight.t())
                    if finishes is not None:
                        fin = _lib.ReduceProblem()
                        _lib.check(lib.bg_mlp_layer_backward_partial(B, C_out, C_in, _lib.ptr(g), _lib.ptr(self.wt[i]), _lib.ptr(a_in), _lib.ptr(self.gin[i]),
                                                                     _lib.ptr(below.bias.grad), _lib.ptr(self.cs[i - 1]), fin, stream),
                                   "bg_mlp_layer_backward_partial")
                        finishes.append(fin)
                    else:
                        _lib.check(lib.bg_mlp_layer_backward(B, C_out, C_in, _lib.ptr(g), _lib.ptr(self.wt[i]), _lib.ptr(a_in), _lib.ptr(self.gin[i]),
                                                             _lib.ptr(below.bias.grad), _lib.ptr(self.cs[i - 1]), stream), "bg_mlp_layer_backward")
                else:
                    torch.mm(g, l.weight, out=self.gin[i])
                    _lib.check(lib.bg_elu_backward_colsum(B, C_in, _lib.ptr(self.gin[i]), _lib.ptr(a_in), _lib.ptr(below.bias.grad), _lib.ptr(self.cs[i - 1]),
                                                          stream), "bg_elu_backward_colsum")
                g = self.gin[i]


class ActorCritic(torch.nn.Module):
    def __init__(self, num_act, num_obs, num_privileged_obs):
        super().__init__()
        self.critic = _mlp(num_obs + num_privileged_obs, CRITIC_HIDDEN, 1)
        self.actor = _mlp(num_obs, ACTOR_HIDDEN, num_act)
        self.logstd = torch.nn.parameter.Parameter(torch.full((1, num_act), fill_value=-2.0), requires_grad=True)

    def act(self, obs):
        mean = self.actor(obs)
        return torch.distributions.Normal(mean, torch.exp(self.logstd).expand_as(mean))

    def est_value(self, obs, privileged_obs):
        return self.critic(torch.cat((obs, privileged_obs), dim=-1)).squeeze(-1)

    # ---- fused rollout inference (reference runner.py:109-111: dist = model.act(obs); act = dist.sample())
    def sample_actions(self, obs, actions_out, seed, counter, mu_out=None):
        if not obs.is_cuda:
            raise RuntimeError("sample_actions runs the fused HIP actor kernel and needs CUDA tensors")
        a = self.actor
        w = [a[0].weight, a[0].bias, a[2].weight, a[2].bias, a[4].weight, a[4].bias, a[6].weight, a[6].bias, self.logstd]
        for t in w + [obs, actions_out]:
            if not t.is_contiguous():
                raise RuntimeError("sample_actions needs contiguous tensors")
        _lib.check(_lib.load().bg_actor_sample(obs.shape[0], _lib.ptr(obs), *[_lib.ptr(t) for t in w], int(seed), int(counter), _lib.ptr(mu_out),
                                               _lib.ptr(actions_out), _lib.current_stream_ptr()), "bg_actor_sample")
        return actions_out
