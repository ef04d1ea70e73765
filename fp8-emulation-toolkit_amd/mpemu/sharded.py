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
import torch
import torch.distributed as dist

from .pytquant.hip import _binding as _hip

# Shards start on a multiple of this many elements: keeps every shard 16-byte aligned for fp32 and
# 16-bit storage and on a noise-superblock boundary of the stochastic modes (2048 = 8 x 64 x 4).
SHARD_ALIGN = 2048


def group_backend(group=None):
    """Backend name of `group` ("nccl", "gloo", ...), or None when torch.distributed is not initialised."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    return str(dist.get_backend(group)).lower()


def all_reduce_max_words(words, group=None, check_equal=None):
    """In-place MAX all-reduce of an int32 device (or CPU) tensor of amax words over `group`.

    nccl: on the device, in stream order.  Other backends: staged through the host (module docstring).
    check_equal: optional Python int that every rank must hold the same value of (the stochastic call
    offset): verified for free on the host-staged path, and on the nccl path when MPEMU_CHECK_OFFSETS=1
    (it costs a host sync there).  Raises RuntimeError when ranks disagree."""
    backend = group_backend(group)
    if backend is None or dist.get_world_size(group) == 1:
        return words
    verify = check_equal is not None and (backend != "nccl" or _CHECK_OFFSETS)
    if verify:
        # max(v) and max(-v) agree in magnitude on every rank iff all ranks hold the same v
        v = int(check_equal) & 0x3FFFFFFF
        extra = torch.tensor([v, -v], dtype=torch.int32, device=words.device)
        buf = torch.cat([words.reshape(-1), extra])
    else:
        buf = words
    if backend == "nccl":
        dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=group)
        host = buf[-2:].cpu() if verify else None
    else:
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


import os as _os

_CHECK_OFFSETS = _os.environ.get("MPEMU_CHECK_OFFSETS", "0") not in ("", "0")


def shard_bounds(numel, world_size, rank, align=SHARD_ALIGN):
    """[start, end) of `rank`'s contiguous shard of a flat tensor of `numel` elements."""
    per = -(-numel // world_size)
    per = -(-per // align) * align
    start = min(numel, rank * per)
    return start, min(numel, start + per)


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

    def cast(self, shard, mode, scale=1.0, elem_base=0, inplace=False, seed=None, offset=None):
        """Unscaled / given-scale cast of the shard: no communication."""
        fmt, rmode = self.ops.parse_mode(mode)
        out = shard if inplace else torch.empty_like(shard)
        self.ops.cast_(shard, out, fmt, rmode, scale=scale, seed=seed, offset=offset, elem_base=elem_base)
        return out

    def amax_words(self, shards):
        """One int32 word per shard (bit pattern of max|x|), reduced over the group in ONE call."""
        dev = shards[0].device
        words = torch.zeros(len(shards), dtype=torch.int32, device=dev)
        for i, s in enumerate(shards):
            self.ops.amax_(s, words[i:i + 1])
        if self._world() > 1:
            all_reduce_max_words(words, self.group)
        return words

    def quantize_per_tensor(self, shard, mode, flt_max, elem_base=0, inplace=False, clamp=True,
                            seed=None, offset=None, amax_word=None):
        """quantize_tensor's per-tensor "max" policy (reference qutils.py:282-287) for a sharded
        tensor: local amax -> all-reduce(MAX) of one word -> device-side scale -> local cast."""
        fmt, rmode = self.ops.parse_mode(mode)
        if amax_word is None:
            amax_word = self.amax_words([shard])
        out = shard if inplace else torch.empty_like(shard)
        self.ops.cast_scaled_(shard, out, fmt, rmode, amax_word, flt_max, clamp=clamp, seed=seed,
                              offset=offset, elem_base=elem_base)
        return out
