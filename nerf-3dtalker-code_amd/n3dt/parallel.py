"""Frame-level data parallelism: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The render path has no cross-frame term, so inference shards frames with no collective at all.
Training replicates the weights and averages gradients.  What is reduced is a *flat gradient arena*: one persistent fp32
buffer per bucket whose slices ARE the parameters' `.grad` tensors (HeadNeRFNet's backward kernels accumulate straight
into its arena, headnerf.py), so a step's collective is `all_reduce(arena)` -- no `cat`, no copy-back, no allocation.
Two buckets in the reference's trainer: HeadNeRFNet (11-14 MB) and the co-trained Audio2style LSTM (86 MB,
talker_trainer.py:428-473, second Adam at :665).  HeadNeRFNet's backward finishes FIRST in `loss.backward()` (its
`audiostyle` input is the LSTM's output, :1008-1063), so its bucket is launched with `async_op=True` from an autograd hook
the moment its last gradient has been accumulated and rides under the LSTM's backward; the second bucket follows at the
end; `GradReducer.wait()` joins both before `optimizer.step()`.  xGMI is point-to-point: a ring all-reduce of 100 MB is
per-link bound (~1.1 ms on one ring); two buckets, not twenty, keep each transfer long enough to fill the links.
"""
import torch
import torch.distributed as dist


def _graph_task_id():
    """Id of the autograd graph task now running (-1 outside backward).  A private torch hook (also used by torch.utils.checkpoint);
    without it a new pass is recognised by the engine callback alone, as before round 4."""
    f = getattr(torch._C, "_current_graph_task_id", None)
    return f() if f is not None else -1


def shard_range(total, rank, world):
    """Contiguous frame range [lo, hi) of `rank`; the first total % world ranks take one extra frame."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class FlatGrads:
    """A persistent flat buffer laid out over a parameter list; `view(i)` is parameter i's gradient slice (every slice
    starts on a 256-byte boundary).  Only parameters that require grad take part; one dtype / device per arena."""

    ALIGN = 64  # elements

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "FlatGrads needs at least one parameter that requires grad"
        p0 = self.params[0]
        assert all(p.dtype == p0.dtype and p.device == p0.device for p in self.params)
        self.offsets, total = [], 0
        for p in self.params:
            self.offsets.append(total)
            total += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.flat = torch.zeros(total, dtype=p0.dtype, device=p0.device)
        self.index = {id(p): i for i, p in enumerate(self.params)}
        self.numel = sum(p.numel() for p in self.params)
        self.touched = False      # a backward handed slices out since the last reduction / zero()
        self._in_pass = False     # slices handed out in the backward pass now running (see hand_out)
        self._pass_id = None      # autograd graph-task id of that pass
        self._handed = set()
        self.in_flight = False    # a collective is reading / writing `flat` (set by GradReducer): nothing may be handed out

    def matches(self, params):
        ps = [p for p in params if p.requires_grad]
        return len(ps) == len(self.params) and all(a is b for a, b in zip(ps, self.params)) and \
            self.flat.device == ps[0].device

    def view(self, i, shape=None):
        """A FRESH tensor object over slice i (autograd adopts a gradient it alone references without copying it)."""
        p = self.params[i]
        return self.flat[self.offsets[i]:self.offsets[i] + p.numel()].view(p.shape if shape is None else shape)

    def is_view(self, i, t):
        return (t is not None and t.untyped_storage().data_ptr() == self.flat.untyped_storage().data_ptr() and
                t.storage_offset() == self.offsets[i] and t.is_contiguous())

    def zero(self):
        self.flat.zero_()
        self.touched = False
        self.end_pass()

    def end_pass(self):
        """Forget the hand-outs of the last backward pass.  The engine's final callback does this when a pass ends normally;
        a backward that RAISED (an out-of-memory batch a training loop skips) never runs it, so hand_out also compares the
        autograd graph-task id, and HeadNeRFNet calls this at the start of every differentiable forward."""
        self._in_pass, self._pass_id = False, None
        self._handed = set()

    def hand_out(self, params):
        """Zeroed gradient slices for `params`, for a backward kernel to accumulate into and return -- or None when the arena
        cannot be used for them in this pass: a parameter already holds a gradient (the caller is accumulating over several
        backward passes: autograd must add, not overwrite), or its slice was already handed out in this pass (the module was
        applied twice in one graph).  The whole arena is zeroed by ONE fill at the first hand-out of a pass."""
        idx = [self.index.get(id(p)) for p in params]
        if any(i is None for i in idx) or self.in_flight:
            return None
        task = _graph_task_id()
        if self._in_pass and task != self._pass_id:
            self.end_pass()  # the pass that set the flag never finished (its backward raised): this is a new one
        if not self._in_pass:
            if any(p.grad is not None for p in self.params):
                return None
            self.flat.zero_()
            self._in_pass, self._pass_id, self._handed = True, task, set()
            # cleared when the engine finishes this backward pass
            torch.autograd.Variable._execution_engine.queue_callback(self._end_pass)
        if any(i in self._handed for i in idx):
            return None
        self._handed.update(idx)
        self.touched = True
        return [self.view(i) for i in idx]

    def _end_pass(self):
        self.end_pass()

    def adopt(self, assign_missing=True):
        """Make every parameter's `.grad` its slice: gradients living elsewhere are copied in (one multi-tensor copy),
        missing ones read as zeros (every rank reduces the same layout).  assign_missing=False leaves the `.grad` of a
        parameter without a gradient None (its slice still reads zero): a collective launched from inside backward must not
        turn a slice it is reducing into a tensor autograd would accumulate a LATE gradient into."""
        src, dst, missing = [], [], []
        for i, p in enumerate(self.params):
            g = p.grad
            if self.is_view(i, g):
                continue
            v = self.view(i)
            if g is None:
                missing.append(v)
                if not assign_missing:
                    continue
            else:
                src.append(g.detach())
                dst.append(v)
            p.grad = v
        if missing and not self.touched:
            # (a pass that handed slices out zeroed the whole arena first: slices nobody wrote are zero already; otherwise
            # they still hold the previous reduction's averages)
            if len(missing) == len(self.params):
                self.flat.zero_()
            else:
                torch._foreach_zero_(missing)
        if src:
            torch._foreach_copy_(dst, src)


_ARENAS = {}


def _arena_for(params):
    """A cached FlatGrads over exactly these parameters (HeadNeRFNet's own arena when the list is its parameter list)."""
    params = [p for p in params if p.requires_grad]
    key = tuple(id(p) for p in params)
    a = _ARENAS.get(key)
    if a is None or not a.matches(params):
        if len(_ARENAS) >= 8:
            _ARENAS.pop(next(iter(_ARENAS)))
        a = _ARENAS[key] = FlatGrads(params)
    return a


def _buckets_of(params):
    """Group a parameter list into reduction buckets: parameters covered by a module-owned arena (HeadNeRFNet.grad_arena())
    keep that arena, the rest share a cached one."""
    owned, rest, seen = [], [], set()
    for p in params:
        if not p.requires_grad:
            continue
        a = getattr(p, "_n3dt_arena", None)
        if a is not None:
            if id(a) not in seen:
                seen.add(id(a))
                owned.append(a)
        else:
            rest.append(p)
    return owned + ([_arena_for(rest)] if rest else [])


def allreduce_gradients(params, world=None, group=None):
    """Average `.grad` over the ranks, in place, one collective per bucket, blocking.  Afterwards every parameter's `.grad`
    is a slice of its bucket's flat buffer; parameters without a gradient contribute zeros.  (GradReducer is the
    overlapping form of the same thing.)"""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    for a in _buckets_of(params):
        a.adopt()
        dist.all_reduce(a.flat, op=dist.ReduceOp.SUM, group=group)
        a.flat.mul_(1.0 / world)
        a.touched = False


class GradReducer:
    """Overlapped gradient averaging over a list of buckets (each a parameter list, or a FlatGrads such as
    `HeadNeRFNet.grad_arena()`), in the order their backward passes finish.

        reducer = GradReducer([net.grad_arena(), audio2style.parameters()])
        ...
        optimizer.zero_grad(); loss.backward(); reducer.wait(); optimizer.step()

    A bucket's collective is launched (async) from a post-accumulate-grad hook as soon as every parameter that received a
    gradient in the PREVIOUS step has received one in this step; whatever was not launched by then (the first step, a step
    whose set of used parameters shrank) is launched by wait().  wait() joins the collectives, scales by 1/world and leaves
    every parameter's `.grad` as a slice of its bucket.

    What happens to a gradient that arrives while its bucket is already being reduced:
      * a parameter that did NOT fire last step (the used set grew: a sub-network was unfrozen, a branch was taken for the
        first time): its slice went out as zeros and its `.grad` was left None, so autograd gives it a fresh tensor, not a
        slice under reduction.  wait() copies such LATE gradients into their slices and reduces those slices in a second
        round of collectives, after the first has completed.  Every rank sees the same late set (the ranks run the same graph).
      * a parameter whose gradient is ALREADY in the bucket (a second backward() before wait(): gradient accumulation; or a
        `.grad` kept alive through zero_grad(set_to_none=False)): autograd accumulates in place into a slice the collective
        is reading and writing.  That cannot be repaired afterwards: wait() raises RuntimeError.  Accumulate under
        `with reducer.no_sync():` (hooks launch nothing there) and run the last backward of the step outside it.

    force=True registers the hooks at world size 1 too (a one-rank process group exercises the same stream ordering between
    the backward kernels and the collective as N ranks do; tests/test_gpu_round4.py)."""

    def __init__(self, buckets, world=None, group=None, force=False):
        self.group = group
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.active = self.world > 1 or force
        self.arenas = [b if isinstance(b, FlatGrads) else _arena_for(list(b)) for b in buckets]
        self.expected = [None] * len(self.arenas)   # ids of the parameters that fired last step
        self.fired = [set() for _ in self.arenas]
        self.late = [dict() for _ in self.arenas]    # id -> parameter that fired after its bucket's launch
        self.work = [None] * len(self.arenas)
        self.launch_order = []
        self.hook_launches = 0   # collectives started from inside backward (the overlapped ones)
        self.late_rounds = 0     # second-round collectives wait() had to run
        self.error = None
        self._sync = True
        self._hooks = []
        if self.active:
            for bi, a in enumerate(self.arenas):
                for p in a.params:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    def _make_hook(self, bi):
        def hook(p):
            if not self._sync:
                return
            f = self.fired[bi]
            if self.work[bi] is not None:
                a = self.arenas[bi]
                if id(p) in f or id(p) in self.late[bi] or a.is_view(a.index[id(p)], p.grad):
                    self.error = ("a gradient was accumulated into bucket %d while its all-reduce was in flight (a second "
                                  "backward() before wait(), or a .grad kept by zero_grad(set_to_none=False)); accumulate "
                                  "under `with reducer.no_sync():`" % bi)
                else:
                    self.late[bi][id(p)] = p
                return
            f.add(id(p))
            exp = self.expected[bi]
            if exp is not None and len(f) == len(exp) and f == exp:
                self._launch(bi, from_hook=True)
                self.hook_launches += 1
        return hook

    def no_sync(self):
        """Context manager for gradient accumulation: backward passes inside it launch no collective and are not counted;
        the step's LAST backward runs outside it, followed by wait()."""
        reducer = self

        class _NoSync:
            def __enter__(self_):
                assert all(w is None for w in reducer.work), "no_sync() entered with a collective in flight: call wait() first"
                reducer._sync = False

            def __exit__(self_, *exc):
                reducer._sync = True
                reducer.fired = [set() for _ in reducer.arenas]  # the closing backward is counted from scratch
                return False
        return _NoSync()

    def _launch(self, bi, from_hook=False):
        a = self.arenas[bi]
        # from inside backward a parameter without a gradient keeps `.grad` None (see the class docstring)
        a.adopt(assign_missing=not from_hook)
        a.in_flight = True
        self.work[bi] = dist.all_reduce(a.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.launch_order.append(bi)

    def wait(self):
        if not self.active:
            return
        for bi in range(len(self.arenas)):
            if self.work[bi] is None:
                self._launch(bi)
        for bi, a in enumerate(self.arenas):
            self.work[bi].wait()
        if self.error is not None:
            msg, self.error = self.error, None
            for bi, a in enumerate(self.arenas):
                self.work[bi], self.fired[bi], self.late[bi], a.in_flight = None, set(), {}, False
            self.launch_order = []
            raise RuntimeError("GradReducer: " + msg)
        for bi, a in enumerate(self.arenas):
            if self.late[bi]:
                # second round: the late gradients move into their (zero) slices and those slices are reduced
                for p in self.late[bi].values():
                    v = a.view(a.index[id(p)])
                    v.copy_(p.grad)
                    p.grad = v
                    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)
                    self.late_rounds += 1
                self.fired[bi] |= set(self.late[bi])
                self.late[bi] = {}
            a.in_flight = False
            a.adopt()  # parameters without a gradient: `.grad` = their (zero) slice, the documented post-condition
            if self.world > 1:
                a.flat.mul_(1.0 / self.world)
            a.touched = False
            self.expected[bi] = self.fired[bi]
            self.fired[bi] = set()
            self.work[bi] = None
        self.last_launch_order, self.launch_order = self.launch_order, []

    def bytes_per_step(self):
        return sum(a.numel * a.flat.element_size() for a in self.arenas)

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def broadcast_parameters(module, src=0, group=None, force=False):
    """Make every rank start from rank `src`'s weights (and buffers): ONE broadcast of a flat buffer, copied back
    into the tensors under no_grad.  The copy moves the version counters (a collective writing a tensor in place does
    not), so packed copies of the weights are rebuilt; modules that keep such copies (HeadNeRFNet.invalidate_packed)
    are told explicitly as well.  force=True broadcasts in a one-rank group too (rehearsals of the N > 1 path)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return
    tensors = list(module.parameters()) + list(module.buffers())
    by_kind = {}
    for t in tensors:
        by_kind.setdefault((t.dtype, t.device), []).append(t)
    with torch.no_grad():
        for ts in by_kind.values():
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
    for m in module.modules():
        if hasattr(m, "invalidate_packed"):
            m.invalidate_packed()


class FlatBucket(torch.nn.Module):
    """A flat fp32 parameter standing in for a co-trained module's gradients in the step's all-reduce.  The reference
    trains an Audio2style LSTM next to the renderer (talker_trainer.py:428-473, second Adam at :665): 21.5 M parameters,
    86 MB of fp32 gradients reduced next to HeadNeRFNet's 11-14 MB."""

    AUDIO2STYLE_PARAMS = 21_546_624  # nn.LSTM(1280, 640, 2 layers, bidirectional) + RNNModel.fc1 (1280->640, unused in forward but a parameter) + Linear 1280-640-320-64 (talker_trainer.py:408-461)

    def __init__(self, numel=AUDIO2STYLE_PARAMS):
        super().__init__()
        self.flat = torch.nn.Parameter(torch.zeros(numel))

    def fill_grad(self, value=0.0):
        if self.flat.grad is None:
            self.flat.grad = torch.full_like(self.flat, value)
        else:
            self.flat.grad.fill_(value)
