"""PPO driver (reference utils/runner.py:19-245, class `Runner`).

Same surface: `Runner(test=False)` parses the reference's 8 CLI flags (runner.py:44-54), merges them over
`envs/<task>.yaml` (:57-68), seeds (:70-80), builds env / model / Adam / buffers (:27-42), `_load` (:82-97),
`train()` (:99-215) and `play()` (:217-241).  What changed is how an iteration executes:

  rollout   per env-step: ONE fused actor+sample launch (bg_actor_sample) and ONE env launch (bg_env_step_to) that writes obs / privileged obs /
            reward / done / time-out directly into rows of the experience buffer -- no `.to(device)` copies, no per-done-env `.item()`
            (runner.py:112-121).  Beside them, on the side stream, the FIRST mini-epoch's forward passes of both networks on each step's rows as soon
            as they exist (same kernels, same weights as the update: bit-identical).
  update    per mini-epoch EIGHT launches on one stream (BG_ONE_STREAM=0: the two networks' chains as separate launches on two streams), no host sync
            inside the 20 mini-epochs (reference: 4 per mini-epoch, runner.py:175,182-184):
              forward        the three hidden layers of BOTH networks as ONE launch that shares the chip by CUs (utils/model.py MLPTrainer ->
                             bg_mlp_chain_forward_split: fp32 operands as exact three-way bf16 splits, all 9 products on the bf16 matrix pipe, fp32
                             accumulation; BG_CHAIN_SPLIT=0: the fp32-MFMA chain bg_mlp_chain_forward_group), the critic's values from the registers
                             of that launch;
              GAE            bg_critic_values_gae: time-out bootstrap, GAE scan, returns, advantage moments in one launch;
              heads + loss   output layers fused with the PPO loss and its backward (bg_critic_head_backward, bg_actor_head);
              backward-data  the hidden layers of both networks as ONE launch (bg_mlp_chain_backward_split: same arithmetic, ELU' and bias-gradient
                             sums in its epilogues; BG_CHAIN_SPLIT_BWD=0: one fp32-MFMA launch per layer);
              weight grads   all six hidden layers of both networks in one grouped launch, the same bf16-pipe arithmetic
                             (bg_mlp_weight_grad_group_split_partial; BG_WGRAD_SPLIT=0: the fp32-MFMA launch bg_mlp_weight_grad_group_partial);
              tail           two launches (bg_update_tail): the deferred fixed-order sums + the squared-norm pieces, then clip + Adam + KL learning-rate
                             rule + statistics bookkeeping + the copies of the weights the layer kernels read (bf16 planes of W, -W, W^T, -W^T).
  multi-GPU one process per GPU (torchrun), environments sharded; per mini-epoch one float64 moments all-reduce on the side stream and ONE grouped RCCL
            launch on the main stream (gradient bucket mean, loss / KL sums, log-std gradient mean) through an own communicator (utils/rccl.py) --
            SURVEY section 8e; the enqueue-order contract of the two communicators is written in utils/parallel.py.

Reference quirks kept on purpose: GAE recomputed from the current critic every mini-epoch (Q5), the
time-out reward overwrite repeated in place (Q4), KL measured with the pre-step distribution (Q6), entropy_coef < 0 (Q7).
"""
import argparse
import ctypes
import glob
import os
import random
import time

import types

import numpy as np
import torch

from .. import _lib
from ..envs import TASKS
from .buffer import ExperienceBuffer
from .config import load_cfg
from .model import ActorCritic, GroupedWeightGrad, MLPTrainer
from .parallel import DataParallel
from .recorder import Recorder
from .utils import (actor_head_forward, actor_head_loss_backward, critic_head_backward, critic_head_forward, critic_values_gae, gae, gaussian_logp, head_scratch, reduce_group,
                    ppo_loss_fused)


def plan_chain_split(slabs_c, slabs_a, cost_c, cost_a, cus, xcds=8):
    """How many of the `cus` one-per-CU persistent workgroups each of the two forward launches of a mini-epoch gets (Runner._plan_chain_split): the
    split that minimises the longer of the two launches, a launch costing ceil(slabs / workgroups) x its slab cost.
    Both counts are multiples of the number of XCDs (8 on MI355X: 32 CUs each).  The hardware deals the workgroups of a launch round-robin over the
    XCDs, so a count that is not a multiple puts one workgroup more on some XCDs; when both launches do that on the same XCD it holds 33 one-per-CU
    workgroups for 32 CUs and the 33rd runs its slabs after a whole persistent workgroup has retired: the launch takes twice as long.  (Seen at
    16,384 envs with 165 + 91: the critic's launch 2,464 us instead of 1,344, 111 ms per iteration instead of 86, in two of three runners built in
    one process -- which XCDs get the extra workgroups depends on the launches before; tools/probe/two_runners.py, HISTORY.md round 5.)"""
    step = xcds if xcds > 0 and cus % xcds == 0 and cus >= 2 * xcds else 1
    best = min(range(step, cus, step), key=lambda a: (max(-(-slabs_c // a) * cost_c, -(-slabs_a // (cus - a)) * cost_a),
                                                       abs(a - cus * slabs_c * cost_c / (slabs_c * cost_c + slabs_a * cost_a))))
    return best, cus - best


class FlatAdam:
    """Adam state on one flat fp32 buffer, stepped by bg_adam_step.  Exposes a torch.optim.Adam-compatible state_dict."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0):
        self.params = list(params)
        dev = self.params[0].device
        self.sizes = [p.numel() for p in self.params]
        # every tensor starts on a 16-byte boundary of the flat buffer (the MFMA actor kernel reads weight rows with 16-byte loads);
        # the padding floats stay zero in params / grads / moments, so norms, Adam and the all-reduce are unaffected
        self.offsets, n = [], 0
        for k in self.sizes:
            self.offsets.append(n)
            n += (k + 3) // 4 * 4
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        for p, k, off in zip(self.params, self.sizes, self.offsets):
            self.flat[off : off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off : off + k].view_as(p)
            p.grad = self.grad[off : off + k].view_as(p)
        self.lr = torch.full((1,), float(lr), dtype=torch.float32, device=dev)
        self.betas, self.eps, self.max_grad_norm = betas, eps, max_grad_norm
        self.step_count = 0
        self._gnorm = torch.zeros(1, dtype=torch.float64, device=dev)

    def zero_grad(self):
        self.grad.zero_()

    def step(self):
        self.step_count += 1
        _lib.check(_lib.load().bg_adam_step(self.flat.numel(), _lib.ptr(self.flat), _lib.ptr(self.grad), _lib.ptr(self.exp_avg),
                                            _lib.ptr(self.exp_avg_sq), _lib.ptr(self.lr), self.step_count, self.betas[0], self.betas[1], self.eps,
                                            self.max_grad_norm, _lib.ptr(self._gnorm), _lib.current_stream_ptr()), "bg_adam_step")

    def step_fused(self, stats, stats_acc, stats_last, kl_index, count, desired_kl, grad_logstd=None, ls_off=0, lr_min=1e-5, lr_max=1e-2, mirrors=None):
        """clip + Adam + KL learning-rate rule + statistics bookkeeping in one launch (bg_optimizer_step): what `step()`, `adapt_lr()` and the
        runner's `stats_acc += stats` / zero fills do as seven dependent launches."""
        self.step_count += 1
        if not hasattr(self, "_ticket"):
            self._ticket = torch.zeros(1, dtype=torch.int32, device=self.flat.device)
        _lib.check(_lib.load().bg_optimizer_step(self.flat.numel(), _lib.ptr(self.flat), _lib.ptr(self.grad), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
                                                 _lib.ptr(self.lr), self.step_count, self.betas[0], self.betas[1], self.eps, self.max_grad_norm,
                                                 _lib.ptr(grad_logstd), int(ls_off), 0 if grad_logstd is None else grad_logstd.numel(), _lib.ptr(stats),
                                                 _lib.ptr(stats_acc), _lib.ptr(stats_last), stats.numel(), int(kl_index), float(count), desired_kl, lr_min,
                                                 lr_max, _lib.ptr(self._ticket), None if mirrors is None else ctypes.addressof(mirrors),
                                                 0 if mirrors is None else len(mirrors), _lib.current_stream_ptr()),
                   "bg_optimizer_step")

    def step_tail(self, wgrad, reductions, stats, stats_acc, stats_last, kl_index, count, desired_kl, grad_logstd=None, ls_off=0, lr_min=1e-5, lr_max=1e-2,
                  mirrors=None):
        """`step_fused` together with the sums in front of it (bg_update_tail: two launches for four): wgrad = (descriptor array, count) of a weight-gradient
        launch made without its finish (GroupedWeightGrad.run(..., finish=False)) or None, reductions = the deferred _lib.ReduceProblem descriptors."""
        self.step_count += 1
        if not hasattr(self, "_tail_sync"):
            self._tail_sync = torch.zeros(4, dtype=torch.int32, device=self.flat.device)
            self._tail_norm = torch.zeros(8192, dtype=torch.float64, device=self.flat.device)
        warr, wn = wgrad if wgrad is not None else (None, 0)
        rarr = (_lib.ReduceProblem * len(reductions))(*reductions) if reductions else None
        _lib.check(_lib.load().bg_update_tail(warr, wn, rarr, len(reductions) if reductions else 0, self.flat.numel(), _lib.ptr(self.flat), _lib.ptr(self.grad),
                                              _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq), _lib.ptr(self.lr), self.step_count, self.betas[0], self.betas[1],
                                              self.eps, self.max_grad_norm, _lib.ptr(grad_logstd), int(ls_off), 0 if grad_logstd is None else grad_logstd.numel(),
                                              _lib.ptr(stats), _lib.ptr(stats_acc), _lib.ptr(stats_last), stats.numel(), int(kl_index), float(count), desired_kl,
                                              lr_min, lr_max, _lib.ptr(self._tail_sync), _lib.ptr(self._tail_norm),
                                              None if mirrors is None else ctypes.addressof(mirrors), 0 if mirrors is None else len(mirrors),
                                              _lib.current_stream_ptr()), "bg_update_tail")

    def tail_sums(self, wgrad, reductions):
        """Launch (1) of `step_tail` alone (bg_update_tail_sums): the ranks of a multi-GPU job average the gradient between the sums and `step_fused`."""
        if not hasattr(self, "_tail_sync"):
            self._tail_sync = torch.zeros(4, dtype=torch.int32, device=self.flat.device)
            self._tail_norm = torch.zeros(8192, dtype=torch.float64, device=self.flat.device)
        warr, wn = wgrad if wgrad is not None else (None, 0)
        rarr = (_lib.ReduceProblem * len(reductions))(*reductions) if reductions else None
        _lib.check(_lib.load().bg_update_tail_sums(warr, wn, rarr, len(reductions) if reductions else 0, _lib.ptr(self._tail_norm), _lib.current_stream_ptr()),
                   "bg_update_tail_sums")

    def adapt_lr(self, kl_sum, count, desired_kl, lr_min=1e-5, lr_max=1e-2):
        _lib.check(_lib.load().bg_adapt_lr(_lib.ptr(kl_sum), float(count), desired_kl, lr_min, lr_max, _lib.ptr(self.lr), _lib.current_stream_ptr()),
                   "bg_adapt_lr")

    def state_dict(self):
        state = {}
        for i, (k, off) in enumerate(zip(self.sizes, self.offsets)):
            state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.exp_avg[off : off + k].view_as(self.params[i]).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off : off + k].view_as(self.params[i]).clone()}
        group = {"lr": float(self.lr.item()), "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(self.sizes)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        for i, (k, off) in enumerate(zip(self.sizes, self.offsets)):
            st = sd["state"].get(i)
            if st is not None:
                self.exp_avg[off : off + k].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off : off + k].copy_(st["exp_avg_sq"].reshape(-1))
                self.step_count = int(float(st["step"]))
        self.lr.fill_(float(sd["param_groups"][0]["lr"]))


class Runner:
    def __init__(self, test=False, args=None, cfg=None):
        self.test = test
        self.dp = DataParallel()
        self.world_size, self.rank, self.local_rank = self.dp.world_size, self.dp.rank, self.dp.local_rank
        if cfg is None:
            self._get_args(args)
            self._update_cfg_from_args()
        else:
            self.cfg = cfg
            self.cfg["basic"].setdefault("task", "T1")
        self.cfg["basic"]["rank"] = self.rank
        if self.world_size > 1:  # one process per GPU: each rank simulates and learns on its own device
            self.cfg["basic"]["sim_device"] = self.cfg["basic"]["rl_device"] = f"cuda:{self.dp.device_index}"
        self._set_seed()
        task = self.cfg["basic"]["task"]
        if task not in TASKS:
            raise NameError(f"name {task!r} is not defined")  # reference: eval(task) (runner.py:27)
        self.env = TASKS[task](self.cfg)

        self.device = self.cfg["basic"]["rl_device"]
        if torch.device(self.device) != torch.device(self.env.device):
            raise ValueError("rl_device must equal sim_device: the rollout writes simulator outputs straight into the PPO buffers")
        self.learning_rate = self.cfg["algorithm"]["learning_rate"]
        self.model = ActorCritic(self.env.num_actions, self.env.num_obs, self.env.num_privileged_obs).to(self.device)
        self.dp.broadcast_parameters(self.model)  # identical initial weights on every rank
        self.invalidate()
        self.optimizer = FlatAdam(self.model.parameters(), lr=self.learning_rate)
        self._load()
        # Resume semantics of the reference (runner.py:31-34,174-180): the restored param_group lr serves the FIRST optimiser step only; the
        # first KL adaptation then overwrites it with a value derived from self.learning_rate = the yaml value, so adaptation restarts from
        # cfg.algorithm.learning_rate, not from the checkpoint's lr.  Same here (see update()).
        self._lr_restart = bool(self.cfg["basic"].get("checkpoint"))
        # common state of the command-curriculum grid across ranks = whatever the env holds now (initial grid, or the restored one)
        self._curr_last = self.env.curriculum_prob.clone() if self.dp.active and self.cfg["commands"].get("curriculum", False) else None

        T, N = self.cfg["runner"]["horizon_length"], self.env.num_envs
        self.buffer = ExperienceBuffer(T, N, self.device)
        self.buffer.add_buffer("actions", (self.env.num_actions,))
        self.buffer.add_buffer("obses", (self.env.num_obs,), extra_rows=1)
        self.buffer.add_buffer("privileged_obses", (self.env.num_privileged_obs,), extra_rows=1)
        self.buffer.add_buffer("rewards", ())
        self.buffer.add_buffer("dones", (), dtype=torch.bool)
        self.buffer.add_buffer("time_outs", (), dtype=torch.bool)
        B, A = T * N, self.env.num_actions
        dev = self.device
        # network inputs with the feature dimension zero-padded to 64 (47 -> 64, 61 -> 64) so that the first layers run on the fused MFMA kernel
        self._pad_in = 64 if MLPTrainer.FUSED else None
        self._critic_in = torch.zeros(T + 1, N, self._pad_in or (self.env.num_obs + self.env.num_privileged_obs), device=dev)
        self._actor_in = torch.zeros(T, N, self._pad_in, device=dev) if self._pad_in else None
        self._adv = torch.zeros(T, N, device=dev)
        self._ret = torch.zeros(T, N, device=dev)
        self._adv_sums = torch.zeros(3, dtype=torch.float64, device=dev)
        self._grad_mu = torch.zeros(B, A, device=dev)
        self._grad_val = torch.zeros(B, device=dev)
        # the float64 sums of a mini-epoch that every rank needs in full: loss / KL statistics [5] and the log-std gradient [A], contiguous so that ONE
        # collective carries both (exchange (3); the log-std gradient then enters the optimiser launch in float64 and not through the fp32 bucket)
        self._sums = torch.zeros(5 + A, dtype=torch.float64, device=dev)
        self._stats, self._grad_logstd = self._sums[:5], self._sums[5:]
        self._stats_acc = torch.zeros(5, dtype=torch.float64, device=dev)
        self._stats_last = torch.zeros(5, dtype=torch.float64, device=dev)  # the last mini-epoch's sums (kl_mean of the log, runner.py:199)
        # False: the mini-epoch tail as separate launches (bg_adam_step, bg_adapt_lr, torch adds / fills): what the first optimiser step after a
        # checkpoint restore runs (see update()); as an attribute for the tests that compare the two forms
        self._fused_opt = True
        # ... and, single process only, the sums in front of it (weight-gradient finish, deferred reductions) as ONE launch that also leaves the squared
        # gradient norm in pieces, so that the optimiser launch need not read the whole gradient in every workgroup (bg_update_tail); False:
        # reduce_group, weight gradients + finish, optimizer_step as separate launches (what the ranks of a multi-GPU job run: the norm is the averaged gradient's)
        self._one_launch_tail = os.environ.get("BG_ONE_LAUNCH_TAIL", "1") == "1"
        self._old_logp = torch.zeros(B, device=dev)
        self._logstd_grad_view = self.model.logstd.grad.view(-1)
        self._logstd_off = (self._logstd_grad_view.data_ptr() - self.optimizer.grad.data_ptr()) // 4  # position of logstd in the flat buffers
        self._actor_tr, self._critic_tr = MLPTrainer(self.model.actor), MLPTrainer(self.model.critic)
        gs = (self.cfg.get("parallel", {}) or {}).get("gemm_split", 0)
        if gs:  # yaml switch for the split-bf16 GEMMs (process-wide, like the environment variable)
            if int(gs) not in (6, 9):
                raise ValueError(f"parallel.gemm_split must be 0, 6 or 9, got {gs!r}")
            MLPTrainer.SPLIT = int(gs)
        self._wgrad_group = GroupedWeightGrad()
        # the critic's stream goes ahead of the actor's in workgroup dispatch: its chain is the longer one and the actor's loss waits for its GAE
        # (update 23.17 -> 23.06 ms in round 2, 21.83-21.91 -> 21.55-21.65 ms in round 3 against equal priorities, tools/archive/ab_prio.sh)
        self._side_stream = torch.cuda.Stream(device=self.device, priority=-1)
        # The small fixed-order reductions behind the head / backward-layer kernels run as ONE launch in front of the weight gradients instead of
        # inside the chains, where each of them waits 20-45 us for a workgroup slot between the other network's resident GEMM workgroups (see
        # update()).  Measured in the loop (tools/ab_env.sh, 4 alternating runs of 20 iterations each): update 21.68-21.91 ms with the one launch in
        # front of the weight gradients (BG_DEFER_FINISH=1, the default), 21.85-21.97 ms with it on the side stream BESIDE them (=2: it delays the
        # one-workgroup-per-CU launch), 21.99-22.04 ms with the finishes inside the chains (=0).
        self._defer_finish = os.environ.get("BG_DEFER_FINISH", "1") in ("1", "2")
        self._defer_serial = os.environ.get("BG_DEFER_FINISH", "1") != "2"
        # critic output layer + GAE in one launch (bg_critic_values_gae: horizons up to 32 steps); otherwise bg_critic_head_forward, a fill and bg_gae
        self._fused_gae = self.cfg["runner"]["horizon_length"] <= 32
        self._chain_values = True  # ... with the values from the chained forward kernel's value head (False: from the stored activations)
        # the two networks' chained forward launches of a mini-epoch share the chip by CUs (each a persistent grid over its slabs, _plan_chain_split);
        # False: one workgroup per slab, dispatched as CUs fall free
        self._split_chain_cus = os.environ.get("BG_SPLIT_CHAIN_CUS", "1") == "1"
        self._split_bwd_chain_cus = os.environ.get("BG_SPLIT_BWD_CHAIN_CUS", "1") == "1"  # ... and the two chained backward launches
        # Both networks' forward chains as ONE launch and both backward chains as ONE launch, the whole mini-epoch on the main stream (round 6): the
        # networks share the chip by CUs inside one grid exactly as the two launches did, and no kernel of the mini-epoch waits for an event of another
        # stream any more (three hand-overs on its critical path, 7-20 us each in a kernel trace).  0: two launches on two streams
        self._one_stream = os.environ.get("BG_ONE_STREAM", "1") == "1"
        self._gae_scratch = torch.zeros(3 * ((self.env.num_envs + 15) // 16) + 1, dtype=torch.float64, device=self.device)

        # fused output layers + loss (bg_head.hip): both networks end in a 128-wide ELU layer, 12 actions / 1 value (utils/model.py); a model of
        # other widths runs the output layers as library GEMMs + bg_ppo_loss (as an attribute for the test that compares the two forms)
        self._fused_head = A == 12 and self.model.actor[-1].in_features == 128 and self.model.critic[-1].in_features == 128
        self._old_mu = torch.zeros(B, A, device=dev)
        self._values_all = torch.zeros(B + N, device=dev)
        self._head_scratch_a, self._head_scratch_c = head_scratch(dev), head_scratch(dev)
        self._act_counter = 0
        # The forward passes of the first mini-epoch run DURING the rollout (see rollout()): 1 = on (default where the chained kernels apply), 0 = off
        self._rollout_forward = os.environ.get("BG_ROLLOUT_FORWARD", "1") == "1" and self._pad_in is not None and N % 128 == 0
        # ... the rows of this many consecutive steps per group of side-stream launches.  Every group costs the main stream one event record (2.9 us, 5.7
        # with a waiter on another queue: tools/event_cost_probe.py), and from four steps per group on the chained launches need more than the 128 CUs
        # the env step leaves idle and slow it down (110 against 102.5 us).  Same box, 20 iterations each, ms per iteration: off 24.78-24.92, one step
        # per group 24.46-24.49, two 24.39-24.41, three 24.42-24.47, four 24.55-24.66, eight 24.58-24.65 (profiles/r04_rollout_forward_ab.txt).
        self._rollout_group = max(1, int(os.environ.get("BG_ROLLOUT_FORWARD_GROUP", "2")))
        self._fwd_ready = False  # rollout() has left the activations / values / old mu of the whole batch in the trainers' buffers
        self.timers = {"rollout": 0.0, "update": 0.0}

    # ------------------------------------------------------------------ config / seed / checkpoint (runner.py:44-97)
    def _get_args(self, args=None):
        parser = argparse.ArgumentParser()
        parser.add_argument("--task", required=True, type=str, help="Name of the task to run.")
        parser.add_argument("--checkpoint", type=str, help="Path of the model checkpoint to load. Overrides config file if provided.")
        parser.add_argument("--num_envs", type=int, help="Number of environments to create. Overrides config file if provided.")
        parser.add_argument("--headless", type=bool, help="Run headless. Overrides config file if provided.")
        parser.add_argument("--sim_device", type=str, help="Device for physics simulation. Overrides config file if provided.")
        parser.add_argument("--rl_device", type=str, help="Device for the RL algorithm. Overrides config file if provided.")
        parser.add_argument("--seed", type=int, help="Random seed. Overrides config file if provided.")
        parser.add_argument("--max_iterations", type=int, help="Maximum number of training iterations. Overrides config file if provided.")
        self.args = parser.parse_args(args)

    def _update_cfg_from_args(self):
        self.cfg = load_cfg(self.args.task)
        for arg, val in vars(self.args).items():
            if val is not None:
                if arg == "num_envs":
                    self.cfg["env"][arg] = val
                else:
                    self.cfg["basic"][arg] = val
        if not self.test:
            self.cfg["viewer"]["record_video"] = False

    def _set_seed(self):
        if self.cfg["basic"]["seed"] == -1:
            # one draw for the whole job: terrain and the logged config.yaml must be the same on every rank (rank offsets are applied
            # where per-rank streams are wanted: T1._cfg_struct and the rollout seed)
            self.cfg["basic"]["seed"] = self.dp.broadcast_int(np.random.randint(0, 10000))
        seed = self.cfg["basic"]["seed"]
        if self.rank == 0:
            print("Setting seed: {}".format(seed))
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        os.environ["PYTHONHASHSEED"] = str(seed)
        torch.cuda.manual_seed_all(seed)

    def _load(self):
        ck = self.cfg["basic"].get("checkpoint")
        if not ck:
            return
        if ck == "-1" or ck == -1:
            ck = sorted(glob.glob(os.path.join("logs", "**/*.pth"), recursive=True), key=os.path.getmtime)[-1]
            self.cfg["basic"]["checkpoint"] = ck
        print("Loading model from {}".format(ck))
        model_dict = torch.load(ck, map_location=self.device, weights_only=True)
        self.model.load_state_dict(model_dict["model"], strict=False)
        self.invalidate()
        try:
            self.env.curriculum_prob = model_dict["curriculum"]
        except Exception as e:
            print(f"Failed to load curriculum: {e}")
        try:
            self.optimizer.load_state_dict(model_dict["optimizer"])
        except Exception as e:
            print(f"Failed to load optimizer: {e}")

    def invalidate(self):
        """Call after changing the model's parameters or the rollout buffers by any means other than this runner's own rollout() / update() (a
        checkpoint or test that loads weights, a tool that edits buffer["obses"] / buffer["actions"]).  The contract between the two phases:
        rollout() may leave the first mini-epoch's forward passes (activations, values, old mu, old log-probabilities of every row) in the trainers'
        buffers and the weight copies that the layer kernels read current, and update() then trusts both without looking; this forgets them, so the
        next update() recomputes everything from the parameters and the buffers as they are."""
        self._fwd_ready = False
        for tr in (getattr(self, "_actor_tr", None), getattr(self, "_critic_tr", None)):
            if tr is not None:
                tr.mirror_fresh = False

    def checkpoint_dict(self):
        return {"model": self.model.state_dict(), "optimizer": self.optimizer.state_dict(), "curriculum": self.env.curriculum_prob}

    # ------------------------------------------------------------------ one PPO iteration
    def rollout(self):
        """runner.py:106-121: horizon_length env steps with sampled actions, outputs written in place.

        While the simulator steps -- one wave per CU on half of the CUs, latency-bound -- the side stream evaluates what the FIRST mini-epoch of the
        update needs and what does not change until the first optimiser step: the critic's hidden layers and values (runner.py:123-125, 132) and the
        actor's hidden layers and old mu (runner.py:126-129) on each step's 4,096 rows as soon as they exist, with the kernels and weights the
        update would use (32 slabs per network and step on the idle CUs: bit-identical to the full-batch launches, asserted in
        tests/test_gpu_ppo.py), and the old log-probabilities one step behind.  update() then starts at the GAE: no old-mu pass, no forward chains
        in mini-epoch 0.  Nothing on the main stream waits for the side stream here; update()'s own hand-overs order the two."""
        buf, T = self.buffer, self.cfg["runner"]["horizon_length"]
        obses, priv = buf["obses"], buf["privileged_obses"]
        seed = int(self.cfg["basic"]["seed"]) + 1000003 * (self.rank + 1)
        ahead = (self._rollout_forward and self._fused_head and self._fused_gae and self._chain_values and not MLPTrainer.SPLIT
                 and self._prepare_rollout_forward())
        main = torch.cuda.current_stream()
        g, start = self._rollout_group, 0
        with torch.no_grad():
            for n in range(T):
                if ahead and n + 1 - start >= g:
                    self._forward_rows(start, n, main)  # rows of steps start .. n: on the side stream, beside this step's launches
                    start = n + 1
                self.model.sample_actions(obses[n], buf["actions"][n], seed, self._act_counter)
                self._act_counter += 1
                self.env.step_to(buf["actions"][n], obses[n + 1], priv[n + 1], buf["rewards"][n], buf["dones"][n], buf["time_outs"][n])
            if ahead:
                self._forward_rows(start, T, main)  # ... and the observation after the last step: the critic's last_values rows
                self._fwd_ready = True

    def _prepare_rollout_forward(self):
        T, N = self.cfg["runner"]["horizon_length"], self.env.num_envs
        ct, at = self._critic_tr, self._actor_tr
        ct.prepare(self._critic_in.reshape((T + 1) * N, -1), train_rows=T * N)
        at.prepare(self._actor_in.reshape(T * N, -1))
        if not (ct._chainable() and at._chainable()):
            return False
        c_out = ct.layers[-1]
        ct.value_head = (c_out.weight.reshape(-1), c_out.bias, self._values_all)
        at.value_head = None
        return True

    def _forward_rows(self, a, b, main):
        """Side stream: pad the observation rows of steps a .. b (b = T: the observation after the last step, critic only) into the network inputs
        and run both networks' chained forward (+ value head, + the actor's output layer = old mu) on them; old log-probabilities of the steps
        whose actions have been sampled by now.  A row block exists once the env step before it has finished: the side stream waits for the main
        stream's work enqueued so far."""
        T, N = self.cfg["runner"]["horizon_length"], self.env.num_envs
        no, npv = self.env.num_obs, self.env.num_privileged_obs
        buf, side = self.buffer, self._side_stream
        ct, at = self._critic_tr, self._actor_tr
        ba = min(b, T - 1)  # last actor step of the range
        side.wait_stream(main)
        with torch.cuda.stream(side), torch.no_grad():
            logstd = self.model.logstd.reshape(-1)
            if a == 0:  # weights may have changed since the last optimiser step by other means (checkpoint, broadcast): six small copies, off the critical path
                ct.refresh_mirrors(); at.refresh_mirrors()
                self._logp_done = 0
            if self._logp_done < a:  # steps of earlier calls: their actions were sampled on the main stream after their rows were enqueued here
                r0, r1 = self._logp_done * N, a * N
                gaussian_logp(self._old_mu[r0:r1], logstd, buf["actions"][self._logp_done : a].reshape(-1, self.env.num_actions), out=self._old_logp[r0:r1])
                self._logp_done = a
            self._critic_in[a : b + 1, :, :no].copy_(buf["obses"][a : b + 1])
            self._critic_in[a : b + 1, :, no : no + npv].copy_(buf["privileged_obses"][a : b + 1])
            jobs = [(ct, a * N, (b + 1 - a) * N)]
            if a <= ba:
                self._actor_in[a : ba + 1, :, :no].copy_(buf["obses"][a : ba + 1])
                jobs.append((at, a * N, (ba + 1 - a) * N))
            MLPTrainer.forward_rows_group(jobs)
            if a <= ba:
                a_out = at.layers[-1]
                actor_head_forward(at.acts[2][a * N : (ba + 1) * N], a_out.weight, a_out.bias, self._old_mu[a * N : (ba + 1) * N])
            if b == T and self._logp_done < T:  # the last call: every action has been sampled
                r0 = self._logp_done * N
                gaussian_logp(self._old_mu[r0:], logstd, buf["actions"][self._logp_done :].reshape(-1, self.env.num_actions), out=self._old_logp[r0:])
                self._logp_done = T

    def update(self):
        """runner.py:123-189: old log-probs, then mini_epochs full-batch optimiser steps.

        Contract with rollout(): when rollout() has run the first mini-epoch's forward passes (`_fwd_ready`), this method starts from them and from the
        weight copies the last optimiser launch wrote -- parameters and rollout buffers must not have been changed in between except through
        `invalidate()` (which `_load` and the initial broadcast call)."""
        u = self._update_begin()
        with torch.no_grad():
            for epoch in range(self.cfg["runner"]["mini_epochs"]):
                # this mini-epoch's hidden activations and values may be the rollout's: same kernels, same weights
                have_fwd = u.ahead and epoch == 0
                if u.one_stream:
                    self._epoch_on_one_stream(u, have_fwd)
                else:
                    self._epoch_critic_forward_and_gae(u, have_fwd)
                    self._epoch_losses_and_backward(u, have_fwd)
                self._epoch_gradients_and_step(u)
        return self._stats_acc

    def _update_begin(self):
        """What update() does once per call: inputs of both networks, old mu / log-std / log-probabilities (runner.py:123-129) unless rollout() left them,
        zeroed accumulators, the CU split of the forward launches.  Returns the namespace the three phases of a mini-epoch share."""
        cfg, buf = self.cfg, self.buffer
        T, N = cfg["runner"]["horizon_length"], self.env.num_envs
        B, A = T * N, self.env.num_actions
        alg = cfg["algorithm"]
        act_flat = buf["actions"].reshape(B, A)
        no, npv = self.env.num_obs, self.env.num_privileged_obs
        # rollout() may have run the first mini-epoch's forward passes already (activations, values and old mu of every row are in place)
        ahead, self._fwd_ready = self._fwd_ready, False
        if not ahead:
            self._actor_tr.mirror_fresh = self._critic_tr.mirror_fresh = False  # weights may have changed outside the loop below (checkpoint, broadcast)
            self._critic_in[:, :, :no].copy_(buf["obses"])
            self._critic_in[:, :, no : no + npv].copy_(buf["privileged_obses"])
            if self._actor_in is not None:
                self._actor_in[:, :, :no].copy_(buf["obses"][:T])
        obs_flat = self._actor_in.reshape(B, -1) if self._actor_in is not None else buf["obses"][:T].reshape(B, -1)
        critic_all = self._critic_in.reshape((T + 1) * N, -1)  # rows [B, B+N) = the observation after the last step (last_values)
        fused_head = self._fused_head
        logstd_flat = self.model.logstd.reshape(-1)
        a_out, c_out = self._actor_tr.layers[-1], self._critic_tr.layers[-1]
        with torch.no_grad():
            # old mu through the same kernels as the mini-epochs: the first ratio is exactly 1
            if ahead:
                old_mu = self._old_mu
            elif fused_head:
                old_mu = actor_head_forward(self._actor_tr.forward_hidden(obs_flat), a_out.weight, a_out.bias, self._old_mu)
            else:
                old_mu = self._actor_tr.forward(obs_flat).clone()
            old_logstd = self.model.logstd.detach().reshape(-1).clone()
            if not ahead:
                gaussian_logp(old_mu, old_logstd, act_flat, out=self._old_logp)
        self._stats_acc.zero_()
        self._stats.zero_()
        self._grad_logstd.zero_()
        self._plan_chain_split(critic_all.shape[0], B)
        # Two HIP streams: the actor and the critic are independent networks, so the HBM-bound elementwise kernels of one overlap
        # the MFMA-bound GEMMs of the other.  side stream = critic forward -> GAE ... critic backward; main stream = actor.
        main = torch.cuda.current_stream()
        side = self._side_stream
        ct, at = self._critic_tr, self._actor_tr
        one_stream = (self._one_stream and fused_head and self._fused_gae and self._chain_values and self._defer_finish and self._defer_serial
                      and not MLPTrainer.SPLIT and ct.chainable_for(critic_all, B) and at.chainable_for(obs_flat) and ct._chain_split_bwd() and at._chain_split_bwd())
        return types.SimpleNamespace(cfg=cfg, buf=buf, T=T, N=N, B=B, A=A, alg=alg, act_flat=act_flat, ahead=ahead, obs_flat=obs_flat, critic_all=critic_all,
                                     fused_head=fused_head, logstd_flat=logstd_flat, a_out=a_out, c_out=c_out, old_mu=old_mu, old_logstd=old_logstd, main=main, side=side,
                                     mirrors=None, one_stream=one_stream)

    def _epoch_critic_forward_and_gae(self, u, have_fwd):
        """Side stream: the critic's forward pass (unless the rollout ran it), values, time-out bootstrap, GAE, returns, advantage moments and their
        exchange (runner.py:132-145).  Leaves hc / values / gae_done in u."""
        cfg, buf, T, N, B, alg, critic_all, fused_head, c_out, main, side = u.cfg, u.buf, u.T, u.N, u.B, u.alg, u.critic_all, u.fused_head, u.c_out, u.main, u.side
        # parameters updated by the previous optimiser step; the loss accumulators (_stats, _grad_logstd) were zeroed by it (before the loop
        # for the first mini-epoch): both heads add into them
        # the chained forward kernel also evaluates the value head, from the registers that hold the last activations: the launch between the
        # critic's forward and the actor's loss then has 400 KB to read instead of 52 MB
        chain_values = fused_head and self._fused_gae and self._chain_values and self._critic_tr.chainable_for(critic_all, B)
        self._critic_tr.value_head = (c_out.weight.reshape(-1), c_out.bias, self._values_all) if chain_values else None
        side.wait_stream(main)
        with torch.cuda.stream(side):
            if fused_head:
                if have_fwd:
                    hc = self._critic_tr.acts[2]
                else:
                    hc = self._critic_tr.forward_hidden(critic_all, train_rows=B)
                if self._fused_gae:
                    # output layer + timeout bootstrap + GAE + returns + advantage moments in ONE launch in front of the actor's loss
                    v_all = critic_values_gae(None if chain_values else hc, c_out.weight, c_out.bias, buf["rewards"], buf["dones"], buf["time_outs"],
                                              alg["gamma"], alg["lam"], self._values_all, self._adv, self._ret, self._adv_sums, self._gae_scratch)
                else:
                    v_all = critic_head_forward(hc, c_out.weight, c_out.bias, self._values_all)
            else:
                v_all = self._critic_tr.forward(critic_all, train_rows=B).squeeze(-1)
            values, last_values = v_all[:B], v_all[B:]
            if not (fused_head and self._fused_gae):
                gae(buf["rewards"], buf["dones"], buf["time_outs"], values.view(T, N), last_values, alg["gamma"], alg["lam"],
                    advantages=self._adv, returns=self._ret, sums=self._adv_sums)
            self.dp.sum_(self._adv_sums, tag="moments")  # exchange (1), on the side stream: hidden under the actor forward
            gae_done = side.record_event()
        u.hc, u.values, u.gae_done = (hc if fused_head else None), values, gae_done

    def _epoch_losses_and_backward(self, u, have_fwd):
        """The actor's forward pass (main stream), both output layers fused with the loss, both backward-data chains on their streams
        (runner.py:147-163).  Leaves defer / fins / fin_c / fin_a in u."""
        buf, B, alg, act_flat, obs_flat, fused_head, logstd_flat, a_out, c_out, old_mu, old_logstd, main, side = u.buf, u.B, u.alg, u.act_flat, u.obs_flat, u.fused_head, u.logstd_flat, u.a_out, u.c_out, u.old_mu, u.old_logstd, u.main, u.side
        hc, values, gae_done = u.hc, u.values, u.gae_done
        fins = fin_c = fin_a = None
        if fused_head:
            # Output layers fused with the loss (bg_head.hip): per network ONE pass over the [B][128] hidden activations gives the
            # output, the loss terms, dL/dz of the hidden layer and the output layer's gradients.  Both heads add into _stats.
            # defer (the default, see __init__): the small fixed-order reductions behind the head kernels and behind every backward layer
            # (output-layer and bias gradients, loss statistics: nothing a chain needs) run as ONE launch on the side stream beside the
            # weight-gradient launch (bg_reduce_group) instead of inside the chains.
            defer = self._defer_finish and not MLPTrainer.SPLIT
            fins = [] if defer else None
            fin_c, fin_a = (_lib.ReduceProblem(), _lib.ReduceProblem()) if defer else (None, None)
            ha = self._actor_tr.acts[2] if have_fwd else self._actor_tr.forward_hidden(obs_flat)
            with torch.cuda.stream(side):
                critic_head_backward(hc[:B], c_out.weight, values, self._ret.view(B), self._critic_tr.hidden_grad, c_out.weight.grad,
                                     c_out.bias.grad, self._critic_tr.layers[-2].bias.grad, self._stats, self._head_scratch_c, finish=fin_c)
                self._critic_tr.backward_hidden(finishes=fins)
            main.wait_event(gae_done)  # advantages and their moments
            actor_head_loss_backward(ha, a_out.weight, a_out.bias, logstd_flat, act_flat, old_mu, old_logstd, self._old_logp,
                                     self._adv.view(B), self._adv_sums, 0.2, alg["bound_coef"], alg["entropy_coef"],
                                     self._actor_tr.hidden_grad, a_out.weight.grad, a_out.bias.grad, self._actor_tr.layers[-2].bias.grad,
                                     self._grad_logstd, self._stats, self._head_scratch_a, finish=fin_a)
            if self.dp.active and not defer:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    self._exchange_sums()  # exchange (3): loss / KL sums, hidden under the backward passes
            self._actor_tr.backward_hidden(finishes=fins)
        else:
            defer = False
            mu = self._actor_tr.forward(obs_flat)
            main.wait_stream(side)
            ppo_loss_fused(mu, logstd_flat, act_flat, old_mu, old_logstd, self._old_logp, self._adv.view(B), self._adv_sums,
                           values, self._ret.view(B), 0.2, alg["bound_coef"], alg["entropy_coef"], self._grad_mu, self._grad_val,
                           self._grad_logstd, self._stats)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self._exchange_sums()  # exchange (3): loss / KL sums, hidden under the backward passes
                self._critic_tr.backward(self._grad_val.view(B, 1))
            self._actor_tr.backward(self._grad_mu)
        u.defer, u.fins, u.fin_c, u.fin_a = defer, fins, fin_c, fin_a

    def _epoch_on_one_stream(self, u, have_fwd):
        """What _epoch_critic_forward_and_gae + _epoch_losses_and_backward do, as one sequence of launches on the main stream: both forward chains in
        one launch (unless the rollout ran them), values + GAE + moments (+ their exchange), the two output layers fused with the loss, both
        backward-data chains in one launch (runner.py:132-163).  Same kernels on the same slabs with the same reduction order as the two-stream form:
        bit-identical results (tests/test_gpu_ppo.py).  Leaves defer / fins / fin_c / fin_a in u."""
        buf, B, alg, act_flat, obs_flat, critic_all, logstd_flat, a_out, c_out, old_mu, old_logstd = u.buf, u.B, u.alg, u.act_flat, u.obs_flat, u.critic_all, u.logstd_flat, u.a_out, u.c_out, u.old_mu, u.old_logstd
        ct, at = self._critic_tr, self._actor_tr
        ct.value_head = (c_out.weight.reshape(-1), c_out.bias, self._values_all)
        if have_fwd:
            u.main.wait_stream(u.side)  # the rollout ran the forward passes on the side stream
            hc, ha = ct.acts[2], at.acts[2]
        else:
            hc, ha = MLPTrainer.forward_hidden_group([(ct, critic_all, B), (at, obs_flat, None)])
        v_all = critic_values_gae(None, c_out.weight, c_out.bias, buf["rewards"], buf["dones"], buf["time_outs"], alg["gamma"], alg["lam"], self._values_all,
                                  self._adv, self._ret, self._adv_sums, self._gae_scratch)
        values = v_all[:B]
        self.dp.sum_(self._adv_sums, tag="moments")  # exchange (1)
        fins, fin_c, fin_a = [], _lib.ReduceProblem(), _lib.ReduceProblem()
        critic_head_backward(hc[:B], c_out.weight, values, self._ret.view(B), ct.hidden_grad, c_out.weight.grad, c_out.bias.grad, ct.layers[-2].bias.grad,
                             self._stats, self._head_scratch_c, finish=fin_c)
        actor_head_loss_backward(ha, a_out.weight, a_out.bias, logstd_flat, act_flat, old_mu, old_logstd, self._old_logp, self._adv.view(B), self._adv_sums, 0.2,
                                 alg["bound_coef"], alg["entropy_coef"], at.hidden_grad, a_out.weight.grad, a_out.bias.grad, at.layers[-2].bias.grad,
                                 self._grad_logstd, self._stats, self._head_scratch_a, finish=fin_a)
        MLPTrainer.backward_hidden_group([ct, at], fins)
        u.hc, u.values, u.gae_done = hc, values, None
        u.defer, u.fins, u.fin_c, u.fin_a = True, fins, fin_c, fin_a

    def _epoch_gradients_and_step(self, u):
        """Deferred reductions, all weight gradients, the exchange of the gradient over the ranks, clip + Adam + KL rule (runner.py:162-180)."""
        cfg, B, alg, main, side = u.cfg, u.B, u.alg, u.main, u.side
        defer, fins, fin_c, fin_a, mirrors = u.defer, u.fins, u.fin_c, u.fin_a, u.mirrors
        fused_tail = self._fused_opt and not self._lr_restart
        # the sums of the tail as one launch behind the main weight-gradient kernel + a lean optimiser launch (single process, see __init__)
        # (its norm is assembled from the sums' own pieces: every gradient element must come out of that launch -- all hidden-layer weight
        # gradients from the grouped kernel, everything else from the deferred reductions)
        # (that is: every hidden layer's backward went through the partial kernel, whose finish carries the bias gradient of the layer below --
        # a layer that fell to the library path wrote its bias gradient outside both lists and the norm would miss it)
        one_tail = (fused_tail and self._one_launch_tail and defer and self._defer_serial and len(fins) + 2 <= 8
                    and all(all(tr.wg_slices[:-1]) for tr in (self._critic_tr, self._actor_tr))
                    and len(fins) == sum(1 if tr._chain_split_bwd() else len(tr.layers) - 2 for tr in (self._critic_tr, self._actor_tr)))
        if not u.one_stream:
            main.wait_stream(side)
        if one_tail:
            pass  # the deferred reductions run inside bg_update_tail
        elif defer and self._defer_serial:  # the deferred reductions as one launch in FRONT of the weight gradients (the default)
            reduce_group([fin_c, fin_a] + fins)
            if self.dp.active:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    self._exchange_sums()  # exchange (3), beside the weight gradients
        elif defer:  # BG_DEFER_FINISH=2: ... (+ what depends on them) on the side stream, beside the weight gradients on the main stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                reduce_group([fin_c, fin_a] + fins)
                if self.dp.active:
                    self._exchange_sums()  # exchange (3)
        # all weight gradients after both backward chains, alone on the GPU: one launch pair for the six hidden layers (shapes outside the
        # kernel's range, or MLPTrainer.FUSED_WGRAD = False: library GEMMs, layer by layer)
        wg_partial = self._wgrad_group.run((self._critic_tr, self._actor_tr), finish=not one_tail)
        if defer and (self.dp.active or not self._defer_serial):
            main.wait_stream(side)
        if one_tail and self.dp.active:
            # ranks: the sums, then ONE collective launch on this stream (gradient bucket: mean; loss / KL sums: sum; log-std gradient: mean -- exchanges
            # (2) and (3) of SURVEY 8(e)), then the optimiser launch, which takes the norm of the averaged gradient
            self.optimizer.tail_sums(wg_partial, [fin_c, fin_a] + fins)
            self.dp.exchange_tail_(self.optimizer.grad, self._stats, self._grad_logstd)
        elif not one_tail:
            self.dp.average_(self.optimizer.grad)  # exchange (2): the one collective on the critical path
        if fused_tail:
            # clip + Adam + KL rule + statistics bookkeeping (and the zeroing of the accumulators for the next mini-epoch) in ONE launch
            # ... and the copies of the weights that the layer kernels read (zero-padded first layers, transposed hidden layers): written by the
            # same launch instead of six strided torch copies inside the chains of the next mini-epoch
            if mirrors is None:
                ms = self._critic_tr.mirror_descriptors(self.optimizer.flat) + self._actor_tr.mirror_descriptors(self.optimizer.flat)
                mirrors = (_lib.ParamMirror * len(ms))(*ms) if 0 < len(ms) <= 16 else None
            if one_tail and not self.dp.active:
                self.optimizer.step_tail(wg_partial, [fin_c, fin_a] + fins, self._stats, self._stats_acc, self._stats_last, 4, B, alg["desired_kl"],
                                         grad_logstd=self._grad_logstd, ls_off=self._logstd_off, mirrors=mirrors)
            else:
                self.optimizer.step_fused(self._stats, self._stats_acc, self._stats_last, 4, B * self.world_size, alg["desired_kl"],
                                          grad_logstd=self._grad_logstd, ls_off=self._logstd_off, mirrors=mirrors)
            self._actor_tr.mirror_fresh = self._critic_tr.mirror_fresh = mirrors is not None
        else:
            self._logstd_grad_view.copy_(self._grad_logstd)  # (behind the bucket's all-reduce, which carries a stale value in this slot)
            self.optimizer.step()
            self._actor_tr.mirror_fresh = self._critic_tr.mirror_fresh = False  # this launch does not write the weight copies: the next pass copies them
            if self._lr_restart:  # first step after a checkpoint load: see __init__
                self.optimizer.lr.fill_(float(cfg["algorithm"]["learning_rate"]))
                self._lr_restart = False
            self.optimizer.adapt_lr(self._stats[4:5], B * self.world_size, alg["desired_kl"])
            self._stats_acc += self._stats
            self._stats_last.copy_(self._stats)
            self._stats.zero_()
            self._grad_logstd.zero_()
        u.mirrors = mirrors

    def _plan_chain_split(self, rows_c, rows_a):
        """The two networks' chains of a mini-epoch run side by side -- inside one grid (the default) or as two launches on two streams -- one workgroup
        per CU (all of its LDS).  Left to the dispatcher, equal-sized slabs of unequal cost run in lockstep rounds and the last 32 slabs run alone (370 us
        for 325 us of work per CU); here each network gets a share of the CUs whose workgroups walk its slabs (bg_mlp_chain_split::workgroups): the
        split that minimises the longer of the two, slab cost ~ flops.  Which workgroup walks which slab changes no bit of the results."""
        ct, at = self._critic_tr, self._actor_tr
        ct.chain_workgroups = at.chain_workgroups = 0
        if not (self._split_chain_cus and ct._chainable() and at._chainable()):
            return
        cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        sc, sa = (rows_c + 127) // 128, (rows_a + 127) // 128
        if sc + sa <= cus:
            return
        cost = lambda tr: sum(l.weight.shape[0] * (tr._kin if i == 0 else l.weight.shape[1]) for i, l in enumerate(tr.layers[:3]))
        ct.chain_workgroups, at.chain_workgroups = plan_chain_split(sc, sa, cost(ct), cost(at), cus)
        fixed = os.environ.get("BG_FWD_CHAIN_CUS")  # "critic,actor": a fixed split (A/B runs)
        if fixed:
            ct.chain_workgroups, at.chain_workgroups = (int(v) for v in fixed.split(","))
        # the chained backward launches likewise (both networks differentiate the same rows_a rows; slab cost ~ flops of the two backward layers)
        ct.chain_bwd_workgroups = at.chain_bwd_workgroups = 0
        if self._split_bwd_chain_cus and ct._chain_split_bwd() and at._chain_split_bwd():
            bcost = lambda tr: sum(l.weight.shape[0] * l.weight.shape[1] for l in tr.layers[1:3])
            ct.chain_bwd_workgroups, at.chain_bwd_workgroups = plan_chain_split(sa, sa, bcost(ct), bcost(at), cus)
            fixed = os.environ.get("BG_BWD_CHAIN_CUS")  # "critic,actor": a fixed split (A/B runs)
            if fixed:
                ct.chain_bwd_workgroups, at.chain_bwd_workgroups = (int(v) for v in fixed.split(","))

    def _exchange_sums(self):
        """Exchange (3), on the current (side) stream: the loss / KL sums and the log-std gradient of all ranks in one float64 all-reduce; the gradient
        is then the mean over ranks like the bucket's."""
        self.dp.sum_(self._sums, tag="stats")
        if self.world_size > 1:
            self._grad_logstd.mul_(1.0 / self.world_size)

    def iteration(self):
        buf, T = self.buffer, self.cfg["runner"]["horizon_length"]
        self.rollout()
        stats = self.update()
        buf.roll()  # carry the last observation into row 0 of the next rollout
        return stats

    def _sync_curriculum(self):
        """Multi-rank command curriculum: sum every rank's increments of the probability grid since the last sync."""
        if not (self.dp.active and self.cfg["commands"].get("curriculum", False)):
            return
        new = self.dp.sync_grid(self.env.curriculum_prob, self._curr_last)
        self.env.curriculum_prob = new
        self._curr_last = new

    def _summarize(self, stats_acc):
        """Host-side means of the loss terms over the mini-epochs (runner.py:182-204); one device->host read (blocking: tests and tools)."""
        s = torch.cat((stats_acc, self._stats_last, self.optimizer.lr.double())).cpu().tolist()
        return self._summary_from(s)

    def _summary_from(self, s):
        T, N = self.cfg["runner"]["horizon_length"], self.env.num_envs
        B, A, E = T * N * self.world_size, self.env.num_actions, self.cfg["runner"]["mini_epochs"]
        self.learning_rate = s[10]
        return {"value_loss": s[0] / (B * E), "actor_loss": s[1] / (B * E), "bound_loss": s[2] / (B * A * E), "entropy": s[3] / (B * E),
                "kl_mean": s[9] / B, "lr": s[10]}

    # ------------------------------------------------------------------ entry points
    # The reference's loop reads several scalars per mini-epoch with .item() (runner.py:175,182-184) and so stalls the GPU twenty times per
    # iteration.  Here an iteration's scalars (loss sums, learning rate, episode statistics, curriculum levels: 45 numbers) are gathered into ONE
    # device vector and copied to pinned host memory without blocking; they are written to the log while the NEXT iteration runs, with their own
    # iteration number.  The host never waits for the GPU inside the loop, so train() runs at the speed of bench.py's timed region (which calls
    # the same train_iteration); the last iteration's scalars are flushed after the loop.
    def begin_training(self, recorder=None):
        self.recorder = recorder if recorder is not None else Recorder(self.cfg, rank=self.rank)
        obs, infos = self.env.reset()
        self.buffer["obses"][0].copy_(obs)
        self.buffer["privileged_obses"][0].copy_(infos["privileged_obs"])
        n = 11 + 4 + _lib.NUM_REWARD_TERMS + 4
        self._log_dev = torch.zeros(n, dtype=torch.float64, device=self.device)
        self._log_host = [torch.zeros(n, dtype=torch.float64).pin_memory() for _ in range(2)]
        self._log_event = [torch.cuda.Event() for _ in range(2)]
        self._log_pending = None
        self.nonfinite_resets_total = 0.0  # over the whole run (the per-iteration accumulator is reset when it is read)

    def _flush_log(self):
        if self._log_pending is None:
            return
        slot, it = self._log_pending
        self._log_event[slot].synchronize()  # recorded one iteration ago: already complete unless the host ran a whole iteration ahead
        s = self._log_host[slot].tolist()
        self._log_pending = None
        summary = self._summary_from(s[:11])
        ne = 4 + _lib.NUM_REWARD_TERMS
        self.recorder.record_episode_statistics(self.env, self.env.reward_names, it, stats=s[11 : 11 + ne])
        self.nonfinite_resets_total += s[11 + ne - 1]
        lv = s[11 + ne :]
        if self.cfg["commands"].get("curriculum", False):
            self.env.mean_lin_vel_level, self.env.mean_ang_vel_level, self.env.max_lin_vel_level, self.env.max_ang_vel_level = lv
        summary.update({"curriculum/mean_lin_vel_level": self.env.mean_lin_vel_level, "curriculum/mean_ang_vel_level": self.env.mean_ang_vel_level,
                        "curriculum/max_lin_vel_level": self.env.max_lin_vel_level, "curriculum/max_ang_vel_level": self.env.max_ang_vel_level})
        self.recorder.record_statistics(summary, it)

    def train_iteration(self, it):
        """One pass of the reference's training loop body (runner.py:103-213): rollout, update, statistics, curriculum exchange, checkpoint."""
        stats = self.iteration()
        d = self._log_dev
        d[0:5].copy_(stats); d[5:10].copy_(self._stats_last); d[10:11].copy_(self.optimizer.lr)
        ne = 4 + _lib.NUM_REWARD_TERMS
        d[11 : 11 + ne].copy_(self.env.episode_stats(reset=True))
        self._sync_curriculum()
        if self.cfg["commands"].get("curriculum", False):
            lin = self.env.get_field("env_curriculum_level_lin").abs().double()
            ang = self.env.get_field("env_curriculum_level_ang").abs().double()
            d[11 + ne :].copy_(torch.stack((lin.mean(), ang.mean(), lin.max(), ang.max())))
        self._flush_log()  # the previous iteration's scalars: their copy finished long ago
        slot = it & 1
        self._log_host[slot].copy_(d, non_blocking=True)
        self._log_event[slot].record()
        self._log_pending = (slot, it)
        if (it + 1) % self.cfg["runner"]["save_interval"] == 0:
            self.recorder.save(self.checkpoint_dict(), it + 1)

    def train(self):
        self.begin_training()
        max_it = self.cfg["basic"]["max_iterations"]
        for it in range(max_it):
            self.train_iteration(it)
            if self.rank == 0:
                print("epoch: {}/{}".format(it + 1, max_it))
        self._flush_log()

    def play(self, max_steps=None, record_path=None):
        """Deterministic rollout with `dist.loc` (runner.py:217-229).  The reference's camera video (runner.py:230-241) is replaced
        by an optional .npz trajectory dump; `max_steps=None` runs until interrupted like the reference."""
        obs, infos = self.env.reset()
        traj, step = [], 0
        try:
            while max_steps is None or step < max_steps:
                with torch.no_grad():
                    act = self.model.actor(obs)
                    obs, rew, done, infos = self.env.step(act)
                if record_path is not None:
                    traj.append({"root": self.env.root_states[0].cpu().numpy(), "dof_pos": self.env.dof_pos[0].cpu().numpy(), "rew": float(rew[0])})
                step += 1
        except KeyboardInterrupt:
            pass
        if record_path is not None and traj:
            np.savez(record_path, root=np.stack([t["root"] for t in traj]), dof_pos=np.stack([t["dof_pos"] for t in traj]),
                     rew=np.array([t["rew"] for t in traj]))
        return step
