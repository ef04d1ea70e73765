"""mpemu.sharded -- the fpemu cast over a tensor sharded across GPUs (one process per GPU).

No counterpart in the reference: `mpemu` has no collective anywhere; its
examples only wrap whole models in DDP.  The cast is elementwise, so a flat tensor shards into
contiguous ranges with NO data-path communication.  Only per-tensor "max" scaling needs an
exchange, and exactly one: each rank reduces |x| over its shard to one 32-bit word on the device,
the words are all-reduced with MAX (RCCL over xGMI when the group's backend is "nccl"), and every
rank derives the same scale on the device.  max is exact and order independent, and the
stochastic-rounding noise is a function of the GLOBAL element index, so an N-GPU result is
bit-identical to the 1-GPU result (tests/test_sharded_gloo.py, tests/test_gpu_sharded.py).

Payload is 4 bytes per tensor: the xGMI links' bandwidth is irrelevant, the cost is one
collective latency.  Callers with many tensors should batch: `amax_words()` fills a vector of
words, one all-reduce covers them all.

Backend rule (`all_reduce_max_words`): a group whose backend is "nccl" (RCCL) reduces the device
words in stream order, no host sync.  Any other backend (gloo: CPU rehearsals, ranks sharing one
GPU) gets the words staged through the host by this module: one synchronising copy out, a CPU
all-reduce, one copy back.  That is what torch's own gloo device-tensor path does internally (a
pinned bounce buffer per call plus a stream synchronise on a gloo worker thread), minus the
per-call pinned allocation and the second thread; see DESIGN.md section 4 for what was measured.
"""
import os

import torch
import torch.distributed as dist

from .pytquant.hip import _binding as _hip

# Shards start on a multiple of this many elements: keeps every shard 16-byte aligned for fp32 and
# 16-bit storage and on a noise-superblock boundary of the stochastic modes (2048 = 8 x 64 x 4).
SHARD_ALIGN = 2048
# A group of one rank has nothing to reduce and all_reduce_max_words returns at once.  tests/test_gpu_sharded.py
# clears this to send the nccl branch (torch -> RCCL all-reduce(MAX) of int32 words, the device-side offset guard)
# through a one-rank RCCL communicator: the only way to execute that branch on a box with a single GPU.
_SINGLE_RANK_SHORTCUT = True
# MPEMU_CHECK_OFFSETS=1: the nccl branch verifies the stochastic call offset synchronously at every call (a host
# sync per call; the default is the device-side guard that later calls poll).
_CHECK_OFFSETS = os.environ.get("MPEMU_CHECK_OFFSETS", "0") not in ("", "0")


def group_backend(group=None):
    """Backend name of `group` ("nccl", "gloo", ...), or None when torch.distributed is not initialised."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    return str(dist.get_backend(group)).lower()


class _OffsetGuard(object):
    """The stochastic call offset must advance in lock-step on every rank (it selects the Philox stream of a call).
    On the nccl path the (o, -o) pair rides along with the amax words through the all-reduce(MAX) and is compared ON
    THE DEVICE into an error word -- no host sync on the hot path.  The word reaches the host through an asynchronous
    copy into pinned memory that later calls poll (event query, never a wait): a disagreement raises a few calls
    after it happened.  check() is the synchronous form (teardown, tests, MPEMU_CHECK_OFFSETS=1)."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.err = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.on_gpu = self.device.type == "cuda"
        self.host = torch.zeros(1, dtype=torch.int32, pin_memory=True) if self.on_gpu else self.err
        self.event = None

    def note(self, pair):
        """pair: int32[2] on the guard's device holding (max over ranks of v, max over ranks of -v)."""
        self.err.bitwise_or_((pair[0:1] + pair[1:2]).ne(0).to(torch.int32))

    def _raise(self):
        self.err.zero_()
        self.host.zero_()
        self.event = None
        raise RuntimeError("mpemu.sharded: ranks disagree on the stochastic call offset: every rank must issue the "
                           "same sequence of stochastic casts")

    def poll(self):
        """Non-blocking: look at the last asynchronous copy if it has landed, then start the next one."""
        if not self.on_gpu:
            if int(self.err[0]) != 0:
                self._raise()
            return
        if torch.cuda.is_current_stream_capturing():
            return
        if self.event is not None:
            if not self.event.query():
                return
            self.event = None
            if int(self.host[0]) != 0:
                self._raise()
        self.host.copy_(self.err, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream(self.device))

    def check(self):
        """Blocking: raise if any call so far saw disagreeing offsets."""
        if int(self.err.cpu()[0]) != 0:
            self._raise()


_guards = {}


def offset_guard(device):
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    g = _guards.get(dev)
    if g is None:
        g = _guards[dev] = _OffsetGuard(dev)
    return g


def check_offsets():
    """Synchronous verification of every device's offset guard (call at teardown or wherever a sync is free)."""
    for g in list(_guards.values()):
        g.check()


def all_reduce_max_words(words, group=None, check_equal=None):
    """In-place MAX all-reduce of an int32 device (or CPU) tensor of amax words over `group`.

    nccl: on the device, in stream order.  Other backends: staged through the host (module docstring).
    check_equal: optional Python int that every rank must hold the same value of (the stochastic call
    offset).  Host-staged path: verified on the spot, for free.  nccl path: the pair (v, -v) is reduced with the
    words and compared on the device into an error word (_OffsetGuard) that later calls poll without blocking;
    MPEMU_CHECK_OFFSETS=1 makes every call check synchronously.  Raises RuntimeError when ranks disagree."""
    backend = group_backend(group)
    if backend is None or (_SINGLE_RANK_SHORTCUT and dist.get_world_size(group) == 1):
        return words
    verify = check_equal is not None
    if verify:
        # max(v) and max(-v) agree in magnitude on every rank iff all ranks hold the same v
        v = int(check_equal) & 0x3FFFFFFF
        extra = torch.tensor([v, -v], dtype=torch.int32, device=words.device)
        buf = torch.cat([words.reshape(-1), extra])
    else:
        buf = words
    if backend == "nccl":
        dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=group)
        if verify:
            words.reshape(-1).copy_(buf[:-2])
            guard = offset_guard(words.device)
            guard.note(buf[-2:])
            if _CHECK_OFFSETS:
                guard.check()
            else:
                guard.poll()
        return words
    host = buf.cpu() if buf.is_cuda else buf
    dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
    if buf.is_cuda:
        buf.copy_(host)
    if verify:
        if int(host[-2]) != -int(host[-1]):
            raise RuntimeError("mpemu.sharded: ranks disagree on the stochastic call offset (max %d, min %d): "
                               "every rank must issue the same sequence of stochastic casts"
                               % (int(host[-2]), -int(host[-1])))
        words.reshape(-1).copy_(buf[:-2])
    return words


def shard_bounds(numel, world_size, rank, align=SHARD_ALIGN):
    """[start, end) of `rank`'s contiguous shard of a flat tensor of `numel` elements."""
    per = -(-numel // world_size)
    per = -(-per // align) * align
    start = min(numel, rank * per)
    return start, min(numel, start + per)


def row_segments(numel, cols, start):
    """The row segments of a contiguous piece [start, start + numel) of a flattened [rows, cols] tensor:
    a list of (local_offset, nrows, ncols, row0, col0) -- at most a partial head row, a run of whole rows and a
    partial tail row (a piece inside one row is a single segment)."""
    segs = []
    pos, end = start, start + numel
    if pos < end and pos % cols:                       # head: the rest of row pos // cols
        stop = min(end, (pos // cols + 1) * cols)
        segs.append((pos - start, 1, stop - pos, pos // cols, pos % cols))
        pos = stop
    whole = (end - pos) // cols
    if whole > 0:
        segs.append((pos - start, whole, cols, pos // cols, 0))
        pos += whole * cols
    if pos < end:                                      # tail: the beginning of the next row
        segs.append((pos - start, 1, end - pos, pos // cols, 0))
    return segs


_LONG_ROW = 1 << 16   # rows longer than this, fewer than 64 of them: full-width launches row by row (as qutils does)


class ShardedCaster(object):
    """Casts this rank's shard of a logical flat tensor.

    group   : torch.distributed process group (None = the default group; may be absent when
              world_size == 1)
    backend : object with parse_mode / cast_ / amax_ / cast_scaled_ (default: the HIP binding).
              Tests inject a CPU stand-in here to exercise the collective logic over gloo.
    """

    def __init__(self, group=None, backend=None):
        self.group = group
        self.ops = backend if backend is not None else _hip

    def _world(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def _collective(self):
        """Whether the words go through all_reduce_max_words (more than one rank; or the tests' one-rank drive)."""
        return self._world() > 1 or (not _SINGLE_RANK_SHORTCUT and group_backend(self.group) is not None)

    def cast(self, shard, mode, scale=1.0, elem_base=0, inplace=False, seed=None, offset=None):
        """Unscaled / given-scale cast of the shard: no communication."""
        fmt, rmode = self.ops.parse_mode(mode)
        out = shard if inplace else torch.empty_like(shard)
        self.ops.cast_(shard, out, fmt, rmode, scale=scale, seed=seed, offset=offset, elem_base=elem_base)
        return out

    def _scratch_word(self, like):
        """One reusable amax word per (device, stream) of this caster: calls on one stream are ordered, so the word of
        the previous call has been consumed by its cast before this call's zeroing runs; no allocation per call."""
        if not like.is_cuda or torch.cuda.is_current_stream_capturing():
            return None
        key = (like.device, torch.cuda.current_stream(like.device).cuda_stream)
        cache = self.__dict__.setdefault("_words", {})
        w = cache.get(key)
        if w is None:
            w = cache[key] = torch.zeros(1, dtype=torch.int32, device=like.device)
        return w

    def amax_words(self, shards, out=None):
        """One int32 word per shard (bit pattern of max|x|), reduced over the group in ONE call.  out: optional int32
        buffer of len(shards) words on the shards' device to use instead of a fresh one."""
        dev = shards[0].device
        if out is not None:
            words = out[:len(shards)]
            words.zero_()
        else:
            words = torch.zeros(len(shards), dtype=torch.int32, device=dev)
        for i, s in enumerate(shards):
            self.ops.amax_(s, words[i:i + 1])
        if self._collective():
            all_reduce_max_words(words, self.group)
        return words

    def quantize_per_tensor_multi(self, shards, mode, flt_max, inplace=True, clamp=True, words=None):
        """Per-tensor "max" quantisation of a LIST of shards (each this rank's piece of its own logical tensor) with
        ONE collective for the whole list: per-tensor amax words in a few launches -> one all-reduce(MAX) of
        len(shards) words -> multi-tensor cast with device-side scales.  Replaces a loop of quantize_per_tensor
        (one collective per tensor; the reference's loop is qutils.py:470-475).  Deterministic modes (the
        multi-tensor cast indexes noise from element 0 of each buffer).  Returns (outputs, words)."""
        shards = list(shards)
        fmt, rmode = mode if isinstance(mode, tuple) else self.ops.parse_mode(mode)
        if rmode == self.ops.STOCHASTIC and self._world() > 1:
            raise NotImplementedError("mpemu.sharded: quantize_per_tensor_multi of sharded tensors is for "
                                      "deterministic rounding modes; use quantize_per_tensor per tensor")
        if not shards:
            return [], words
        if words is None:
            words = torch.empty(len(shards), dtype=torch.int32, device=shards[0].device)
        self.ops.amax_multi_(shards, words)
        if self._collective():
            all_reduce_max_words(words, self.group)
        outs = shards if inplace else [torch.empty_like(t) for t in shards]
        self.ops.cast_multi_scaled_(shards, fmt, rmode, words, flt_max, dsts=outs, clamp=clamp)
        return outs, words

    def quantize_rows_flat(self, shard, rows, cols, start, mode, flt_max, clamp=True, inplace=False, seed=None,
                           offset=None, out=None):
        """quantize_tensor's per-channel "max" policy (reference qutils.py:289-311; clamp=False: the fine-grained
        one, :313-337) for this rank's contiguous piece [start, start + shard.numel()) of a flattened [rows, cols]
        tensor whose rows may straddle ranks: per-row-segment maxima into a vector of `rows` words -> ONE
        all-reduce(MAX) of that vector -> every segment cast with its row's scale.  Stochastic modes: row r draws as
        call `offset + r` and column c as element c of that call, whatever the cut, so the result is the
        single-device result bit for bit; the call consumes `rows` offsets on every rank.
        out: optional contiguous result buffer (may be `shard` itself); returns (result, row amax words)."""
        fmt, rmode = self.ops.parse_mode(mode)
        n = shard.numel()
        if start < 0 or start + n > rows * cols:
            raise ValueError("mpemu.sharded: piece [%d, %d) lies outside the %d x %d tensor" % (start, start + n, rows, cols))
        if not shard.is_contiguous():
            raise ValueError("mpemu.sharded: the piece must be contiguous")
        flat = shard.view(-1)
        result = shard if (inplace or out is shard) else (out if out is not None else torch.empty_like(shard))
        out = result.view(-1)
        if offset is None:
            offset = self.ops.take_offsets(rows) if rmode == self.ops.STOCHASTIC else 0
        words = torch.zeros(rows, dtype=torch.int32, device=shard.device)   # rows this rank holds nothing of stay 0
        segs = row_segments(n, cols, start)
        for lo, nr, nc, r0, c0 in segs:
            piece = flat[lo:lo + nr * nc]
            if nr < 64 and nc > _LONG_ROW:
                for r in range(nr):
                    self.ops.amax_(piece[r * nc:(r + 1) * nc], words[r0 + r:r0 + r + 1])
            else:
                self.ops.amax_rows_(piece, nr, words[r0:r0 + nr])
        if self._collective():
            all_reduce_max_words(words, self.group, check_equal=offset if rmode == self.ops.STOCHASTIC else None)
        for lo, nr, nc, r0, c0 in segs:
            src, dst = flat[lo:lo + nr * nc], out[lo:lo + nr * nc]
            if nr < 64 and nc > _LONG_ROW:
                for r in range(nr):
                    self.ops.cast_scaled_(src[r * nc:(r + 1) * nc], dst[r * nc:(r + 1) * nc], fmt, rmode,
                                          words[r0 + r:r0 + r + 1], flt_max, clamp=clamp, seed=seed,
                                          offset=offset + r0 + r, elem_base=c0)
            else:
                self.ops.cast_rows_amax_(src, dst, nr, fmt, rmode, words[r0:r0 + nr], flt_max, clamp=clamp, seed=seed,
                                         offset=offset + r0, col_base=c0)
        return result, words

    def quantize_per_tensor(self, shard, mode, flt_max, elem_base=0, inplace=False, clamp=True,
                            seed=None, offset=None, amax_word=None):
        """quantize_tensor's per-tensor "max" policy (reference qutils.py:282-287) for a sharded
        tensor: local amax -> all-reduce(MAX) of one word -> device-side scale -> local cast."""
        fmt, rmode = self.ops.parse_mode(mode)
        if amax_word is None:
            amax_word = self.amax_words([shard], out=self._scratch_word(shard))
        out = shard if inplace else torch.empty_like(shard)
        self.ops.cast_scaled_(shard, out, fmt, rmode, amax_word, flt_max, clamp=clamp, seed=seed,
                              offset=offset, elem_base=elem_base)
        return out
