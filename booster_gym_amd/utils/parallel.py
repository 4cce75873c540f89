"""Data-parallel plumbing: one process per GPU, environments sharded across ranks.

The reference is single-process (no torch.distributed anywhere).  Environments are independent, so the only exchange is per
optimiser step: (1) [sum adv, sum adv^2, count] so that every rank normalises advantages with the global moments
(runner.py:145), (2) one flat fp32 gradient bucket (177,945 floats = 712 kB, latency-bound on xGMI: one collective, no
per-parameter hooks), averaged, (3) the loss / KL sums so that every rank takes the same learning-rate branch
(runner.py:174-180).  With equal shards this is algebraically the reference update on the union of all shards.

Backend: "nccl" (= RCCL on ROCm) on GPUs; `BG_DIST_BACKEND=gloo` lets the CPU tests run the same code path; `BG_OWN_RCCL=0` keeps the per-mini-epoch
exchanges on torch.distributed's own communicator (utils/rccl.py otherwise).  `BG_DIST_FORCE=1` takes the
collective path with a world of ONE process too (process group initialised, every exchange issued): the one-GPU box's way to execute the RCCL
calls of this file (tests/test_gpu_rccl.py).

Enqueue-order contract (two communicators, two streams).  A communicator executes its collectives in the order they were ENQUEUED, per rank; ranks
that enqueue the same communicator's collectives in different orders dead-lock or mix up buffers.  This build drives
  * the OWN communicator (utils/rccl.py): tag "moments" (the advantage moments, once per mini-epoch) and tag "bucket" (the grouped gradient /
    statistics / log-std exchange, once per mini-epoch, inside Runner._epoch_gradients_and_step).  In the default one-stream mini-epoch
    (Runner._epoch_on_one_stream) both are enqueued on the MAIN stream: one communicator, one stream.  With BG_ONE_STREAM=0 "moments" is enqueued on
    the side stream (Runner._epoch_critic_forward_and_gae) and "bucket" on the main stream: one communicator driven from two streams of a rank;
  * the PROCESS GROUP's communicator for everything outside the mini-epochs (seed, initial weights, curriculum grid, barriers).
The host enqueues strictly in program order -- moments(e), bucket(e), moments(e + 1), ... -- on every rank, whatever the streams do on the device, and
nothing else touches the own communicator; the process group's collectives are issued only between iterations.  Any change that makes the ORDER OF
HOST CALLS depend on rank-local data (an early exit, a rank-dependent branch around an exchange) breaks the contract.  `BG_DP_LOG_ORDER=1` records the
sequence of (tag, stream) per rank (DataParallel.order_log); tests/test_host_logic.py compares it across the ranks of a two-process job.
"""
import os
import sys
import threading

import torch
import torch.distributed as dist

from .rccl import NCCL_AVG, NCCL_SUM


def own_comm_bring_up(rank, world_size, device_index, ok_device, prepare, finish, timeout_s=None):
    """The ranks' agreement on the own communicator, written so that NO failure on one rank can desynchronise the process group's collectives:

        1. local, cannot block: `prepare(rank)` -> (handle, rank 0's unique id or None); an exception is kept, not raised;
        2. ALWAYS, on every rank: broadcast_object_list of rank 0's id (None if its prepare failed), then a MIN all-reduce of "my prepare worked and I
           hold an id" -- the same two collectives in the same order whatever happened in step 1;
        3. only if every rank said yes: `finish(handle, id, rank, world_size, device_index)` = ncclCommInitRank, which blocks until all ranks are in
           it.  A watchdog thread bounds the wait: after `timeout_s` (BG_RCCL_INIT_TIMEOUT, default 120) it prints the rank and ends the process with
           exit code 3 (os._exit: no re-exec, no hang -- the launcher sees a failed rank);
        4. ALWAYS when step 3 was entered: a second MIN all-reduce of "my communicator came up", so that one rank's quick failure inside step 3
           turns into the fallback everywhere (the ranks still blocked in ncclCommInitRank are ended by their watchdogs).

    Returns (communicator or None, the local exception or None).  None = every rank uses the process group's collectives."""
    timeout_s = float(os.environ.get("BG_RCCL_INIT_TIMEOUT", "120")) if timeout_s is None else float(timeout_s)
    handle, raw, err = None, None, None
    try:
        handle, raw = prepare(rank)
    except Exception as ex:  # (library without the C entry points, ncclGetUniqueId failed, ...)
        err = ex
    box = [raw if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    raw = box[0]
    ok = torch.tensor([1 if (err is None and raw is not None) else 0], dtype=torch.int32, device=ok_device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) != 1:
        return None, err
    comm = None
    done = threading.Event()

    def watchdog():
        if not done.wait(timeout_s):
            sys.stderr.write(f"[booster_gym_amd] rank {rank}: ncclCommInitRank has not returned after {timeout_s:.0f} s (BG_RCCL_INIT_TIMEOUT); "
                             "ending this process with exit code 3\n")
            sys.stderr.flush()
            os._exit(3)

    t = threading.Thread(target=watchdog, daemon=True)
    t.start()
    try:
        comm = finish(handle, raw, rank, world_size, device_index)
    except Exception as ex:
        err = ex
    done.set()
    ok = torch.tensor([0 if comm is None else 1], dtype=torch.int32, device=ok_device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) != 1:
        if comm is not None:
            comm.destroy()
        return None, err
    return comm, None


class DataParallel:
    def __init__(self, backend=None):
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = os.environ.get("BG_DIST_BACKEND", backend or "nccl")
        self.owns_group = False
        self.force = os.environ.get("BG_DIST_FORCE", "0") == "1"
        # bench.py: {"moments": [], "bucket": [], "stats": []} -- every exchange of that kind is then bracketed by HIP events on the stream it is issued on
        self.timed_events = None
        # device of this rank: LOCAL_RANK, unless the launcher already narrowed the visible devices to one per process
        # (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES), or a test shares one GPU between ranks (BG_LOCAL_DEVICE)
        ndev = torch.cuda.device_count()
        self.device_index = int(os.environ.get("BG_LOCAL_DEVICE", self.local_rank if self.local_rank < max(ndev, 1) else 0))
        if (self.world_size > 1 or self.force) and not dist.is_initialized():
            if self.backend == "nccl":
                torch.cuda.set_device(self.device_index)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if self.backend == "nccl":
                # A process group's streams beside this build's two want another mapping of HIP streams onto hardware queues than ROCm's default of 4 per
                # process (50-65 % of the loop's speed).  booster_gym_amd/__init__.py sets GPU_MAX_HW_QUEUES=2 at import for processes
                # that will join a group; the variable is read when the HIP runtime starts, so say so if that could not have worked.
                import warnings

                import booster_gym_amd

                q = os.environ.get("GPU_MAX_HW_QUEUES")
                if booster_gym_amd.HW_QUEUES_SET_TOO_LATE:
                    warnings.warn("GPU_MAX_HW_QUEUES=2 was set after this process had started the HIP runtime and has no effect; import booster_gym_amd (or "
                                  "export the variable) before the first CUDA call")
                elif q is None or not q.isdigit() or not 1 <= int(q) <= 3:  # (best effort: only runtimes started through torch are noticed)
                    warnings.warn(f"GPU_MAX_HW_QUEUES={q!r}: with RCCL's streams beside the update's two, anything but 1-3 hardware queues per process cost "
                                  "40-65 % of the training loop on MI355X (measured); export GPU_MAX_HW_QUEUES=2 before the process starts")
            dist.init_process_group(backend=self.backend, rank=self.rank, world_size=self.world_size)
            self.owns_group = True
        # the per-mini-epoch exchanges go through an own RCCL communicator, issued on the stream of the kernels around them (utils/rccl.py: the process
        # group's collectives run on its own stream between two event hand-overs, +54 us per mini-epoch in a world of one); gloo (CPU tests): the group
        self.comm = None
        if self.active and self.backend == "nccl" and os.environ.get("BG_OWN_RCCL", "1") != "0":  # BG_OWN_RCCL=0: the process group's collectives
            from .rccl import RcclComm

            # every rank takes the same path: the own communicator only if it came up EVERYWHERE, otherwise the process group's collectives (still RCCL,
            # on the group's stream) -- loudly
            self.comm, err = own_comm_bring_up(self.rank, self.world_size, self.device_index, f"cuda:{self.device_index}", RcclComm.prepare, RcclComm)
            if self.comm is None:
                import warnings

                warnings.warn(f"own RCCL communicator unavailable on at least one rank ({err!r} here): the per-mini-epoch exchanges go through torch.distributed's "
                              "NCCL backend (+0.5 ms per iteration of stream hand-overs, DESIGN.md section 8)")
        # BG_DP_LOG_ORDER=1: every collective this object issues is noted as (tag, stream kind); tests compare the sequences of the ranks (see the
        # enqueue-order contract in the module docstring)
        self.order_log = [] if os.environ.get("BG_DP_LOG_ORDER", "0") == "1" else None

    @property
    def active(self):
        return self.world_size > 1 or self.force

    def _note(self, tag):
        if self.order_log is not None:
            main = torch.cuda.is_available() and torch.cuda.current_stream() == torch.cuda.default_stream()
            self.order_log.append((tag or "untagged", "main" if main or not torch.cuda.is_available() else "side"))

    def _timed(self, tag, t):
        ev = self.timed_events
        if ev is None or tag not in ev or not t.is_cuda:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ev[tag].append((e0, e1))
        return e1

    def sum_(self, t, tag=None):
        """In-place SUM all-reduce (no-op for a single process).  tag: which exchange this is, for bench.py's per-exchange timing."""
        if self.active:
            self._note(tag)
            e1 = self._timed(tag, t)
            if self.comm is not None and t.is_cuda:
                self.comm.all_reduce_(t, NCCL_SUM)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
            if e1 is not None:
                e1.record()
        return t

    def average_(self, t):
        """In-place mean over ranks: the flat gradient bucket.  With `timed_events` armed (bench.py) every call is bracketed by events on the
        current stream: the span is the collective plus the wait for the slowest rank to arrive."""
        if self.active:
            self._note("bucket")
            e1 = self._timed("bucket", t)
            if self.comm is not None and t.is_cuda:  # RCCL averages inside the collective: no second launch on the critical path of every mini-epoch
                self.comm.all_reduce_(t, NCCL_AVG)
            else:                       # gloo has no AVG
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                t.mul_(1.0 / self.world_size)
            if e1 is not None:
                e1.record()
        return t

    def exchange_tail_(self, bucket, stats, grad_logstd):
        """The exchanges behind the gradient's last sums as ONE collective launch on the current stream: the flat gradient bucket (mean over ranks), the
        loss / KL sums (sum) and the log-std gradient (mean) -- exchanges (2) and (3) of SURVEY 8(e).  Timed as "bucket" when bench.py arms the events."""
        if not self.active:
            return
        self._note("bucket")
        e1 = self._timed("bucket", bucket)
        if self.comm is not None and bucket.is_cuda:
            with self.comm.group() as c:
                c.all_reduce_(bucket, NCCL_AVG)
                c.all_reduce_(stats, NCCL_SUM)
                c.all_reduce_(grad_logstd, NCCL_AVG)
        else:
            for t in (bucket, stats, grad_logstd):
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
            bucket.mul_(1.0 / self.world_size)
            grad_logstd.mul_(1.0 / self.world_size)
        if e1 is not None:
            e1.record()

    def sync_grid(self, cur, last):
        """Command-curriculum grid under data parallelism (reference envs/t1.py:404-413 on one process): every rank has added
        its own increments to `cur` since the common state `last`; the new common state is last + SUM over ranks of (cur - last), clamped
        at 1 like the reference's update.  `last` must be the grid all ranks shared at the previous sync -- after a checkpoint restore that
        is the RESTORED grid, not the initial one (otherwise the restored part would be counted world_size times)."""
        delta = cur - last
        self.sum_(delta)
        return torch.clamp(last + delta, max=1.0)

    def broadcast_int(self, value, src=0):
        """One Python int from rank `src` to every rank (the drawn seed when basic.seed == -1)."""
        if not self.active:
            return int(value)
        box = [int(value)]
        dist.broadcast_object_list(box, src=src)
        return int(box[0])

    def max_(self, t):
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t

    def broadcast_parameters(self, module, src=0):
        if self.active:
            for p in module.parameters():
                dist.broadcast(p.data, src=src)

    def barrier(self):
        if self.active:
            if self.backend == "nccl":  # name the device: without it the backend guesses one from the global rank
                dist.barrier(device_ids=[self.device_index])
            else:
                dist.barrier()

    def shutdown(self):
        if self.comm is not None:
            torch.cuda.synchronize()
            self.comm.destroy()
            self.comm = None
        if self.active and self.owns_group and dist.is_initialized():
            dist.destroy_process_group()
