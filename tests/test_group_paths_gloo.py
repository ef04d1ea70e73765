"""The group-aware paths beyond the single per-tensor cast, rehearsed on CPU:
world_size-2 gloo processes run mpemu.sharded / mpemu.qutils with the oracle-backed stand-in for the HIP ops and
must reproduce the single-process result bit for bit.
  * per-channel / fine-grained "max" scaling of a FLAT shard whose rows straddle ranks: one all-reduce(MAX) of the
    vector of row maxima (reference loop: qutils.py:289-337);
  * quantize_model_weights under a group: one collective for all weights (reference loop: qutils.py:470-475);
  * the stochastic-offset guard that verifies lock-step offsets on the device, without a host sync."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(rank, world, port):
    for p in (os.path.join(ROOT, "fp8-emulation-toolkit_amd"), os.path.join(ROOT, "oracle"),
              os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _matrix(rows, cols, seed=41):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((rows, cols)) * np.exp(rng.standard_normal((rows, 1)) * 2.0)).astype(np.float32)
    x[1] = 0.0                      # an all-zero row: vmax 0 -> inf -> clamp
    x[rows - 1, cols - 1] = 777.0   # the last rank holds the last row's maximum
    x[rows // 2, cols - 1] = -555.0
    return x


def _rows_oracle(x, mode, flt_max, clamp=True, seed=0, offset=0):
    import orc

    out = np.empty_like(x)
    for r in range(x.shape[0]):
        s = orc.scale_from_amax(orc.amax(x[r]), flt_max, clamp=clamp)
        out[r] = orc.cast(x[r], mode, scale=s, seed=seed, offset=offset + r)
    return out


def _cut(n, world, rank, align):
    """Contiguous pieces of a flat tensor on a multiple of `align` elements (the last one takes the rest)."""
    per = -(-n // world)
    per = -(-per // align) * align
    a = min(n, rank * per)
    return a, min(n, a + per)


def _flat_rows_worker(rank, world, port, rows, cols, align, outdir):
    _setup(rank, world, port)
    import _cpu_backend
    from mpemu import qutils
    from mpemu.sharded import ShardedCaster

    x = _matrix(rows, cols).reshape(-1)
    a, b = _cut(x.size, world, rank, align)
    shard = torch.from_numpy(x[a:b].copy())
    sc = ShardedCaster(backend=_cpu_backend)
    res = {}
    res["rne"], res["words"] = sc.quantize_rows_flat(shard, rows, cols, a, "E4M3_RNE", 448.0)
    _cpu_backend.set_rng_state((0, 11))
    res["sr"], _ = sc.quantize_rows_flat(shard, rows, cols, a, "E5M2_STOCHASTIC", 57344.0)      # offsets 11 .. 11+rows
    res["sr_next"] = torch.tensor(_cpu_backend.get_rng_state()[1])
    res["noclamp"], _ = sc.quantize_rows_flat(shard, rows, cols, a, "E3M4_RNE", 30.0, clamp=False)
    # the hooks' entry point: quantize_tensor with a per-channel configuration on the flat shard
    qutils._hip = _cpu_backend
    qutils._device_tensor = lambda t: True   # the stand-in works on CPU tensors
    qutils.set_amax_process_group(dist.group.WORLD, elem_base=a, layout="flat", logical=(rows, cols))
    res["qt"] = qutils.quantize_tensor(shard, qutils.TensorQuantConfig("e4m3", "rne", "per-channel"), None)
    if cols % 8 == 0:
        res["fine"] = qutils.quantize_tensor(shard, qutils.TensorQuantConfig("e4m3", "rne", "fine-grained", group_size=8), None)
    qutils.set_amax_process_group(dist.group.WORLD, elem_base=a)        # no layout: refused
    try:
        qutils.quantize_tensor(shard, qutils.TensorQuantConfig("e4m3", "rne", "per-channel"), None)
        res["refused"] = torch.tensor(0)
    except RuntimeError as e:
        res["refused"] = torch.tensor(int("layout" in str(e)))
    qutils.set_amax_process_group(None)
    torch.save(res, os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("rows,cols,align", [(13, 40, 16), (6, 24, 8), (3, 1000, 16), (40, 7, 1)])
def test_flat_shard_per_channel_equals_single_process(tmp_path, rows, cols, align):
    from _util import assert_bits_equal

    world = 2
    mp.spawn(_flat_rows_worker, args=(world, _free_port(), rows, cols, align, str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=True) for r in range(world)]
    x = _matrix(rows, cols)
    cat = lambda k: np.concatenate([p[k].numpy().reshape(-1) for p in parts]).reshape(rows, cols)
    assert_bits_equal(cat("rne"), _rows_oracle(x, "E4M3_RNE", 448.0), "flat shard, per-channel RNE")
    assert_bits_equal(cat("sr"), _rows_oracle(x, "E5M2_STOCHASTIC", 57344.0, seed=0, offset=11),
                      "flat shard, per-channel SR: row r = call offset + r, noise by column")
    assert_bits_equal(cat("noclamp"), _rows_oracle(x, "E3M4_RNE", 30.0, clamp=False), "flat shard, no clamp")
    assert_bits_equal(cat("qt"), _rows_oracle(x, "E4M3_RNE", 448.0), "quantize_tensor, layout flat")
    if cols % 8 == 0:
        fine = _rows_oracle(x.reshape(-1, 8), "E4M3_RNE", 448.0, clamp=False).reshape(rows, cols)
        assert_bits_equal(cat("fine"), fine, "quantize_tensor fine-grained, layout flat")
    assert all(int(p["sr_next"]) == 11 + rows for p in parts), "every rank consumed `rows` offsets"
    assert all(int(p["refused"]) == 1 for p in parts)
    # the reduced row maxima are the whole tensor's, on every rank
    want = np.abs(x).max(axis=1).astype(np.float32)
    for p in parts:
        assert np.array_equal(p["words"].numpy().view(np.float32), want)


class _Tiny(torch.nn.Module):
    def __init__(self, shard_rows):
        super().__init__()
        self.a = torch.nn.Linear(24, shard_rows)
        self.b = torch.nn.Linear(16, shard_rows)
        self.c = torch.nn.Conv2d(4, shard_rows, 3)
        self.tied = torch.nn.Linear(24, shard_rows)
        self.tied.weight = self.a.weight          # one storage, two modules


def _logical_weights(total_rows):
    rng = np.random.default_rng(5)
    w = {"a": rng.standard_normal((total_rows, 24)), "b": rng.standard_normal((total_rows, 16)) * 3.0,
         "c": rng.standard_normal((total_rows, 4, 3, 3)) * 0.1}
    w = {k: v.astype(np.float32) for k, v in w.items()}
    w["a"][total_rows - 1, 3] = 9.5      # every weight's maximum lives on the last rank
    w["b"][total_rows - 1, 0] = -40.0
    w["c"][total_rows - 1, 0, 0, 0] = 2.25
    return w


def _weights_worker(rank, world, port, outdir):
    _setup(rank, world, port)
    import _cpu_backend
    from mpemu import qutils

    qutils._hip = _cpu_backend
    rows = 6
    w = _logical_weights(world * rows)
    m = _Tiny(rows)
    with torch.no_grad():
        for k in ("a", "b", "c"):
            getattr(m, k).weight.copy_(torch.from_numpy(w[k][rank * rows:(rank + 1) * rows]))
    qutils._device_tensor = lambda t: True   # the stand-in works on CPU tensors
    calls = {"n": 0}
    real = dist.all_reduce

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    dist.all_reduce = counting
    try:
        cfg = qutils.ModuleQuantConfig(wt_qconfig=qutils.TensorQuantConfig("e4m3", "rne", "per-tensor"))
        d = {n: cfg for n in ("a", "b", "c", "tied")}
        qutils.set_amax_process_group(dist.group.WORLD)
        qutils.quantize_model_weights(m, d)
        qutils.set_amax_process_group(None)
    finally:
        dist.all_reduce = real
    res = {k: getattr(m, k).weight.detach().clone() for k in ("a", "b", "c")}
    res["collectives"] = torch.tensor(calls["n"])
    torch.save(res, os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_quantize_model_weights_under_a_group_one_collective(tmp_path):
    import orc
    from _util import assert_bits_equal

    world = 2
    mp.spawn(_weights_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=True) for r in range(world)]
    w = _logical_weights(world * 6)
    q = lambda x: orc.cast(x, "E4M3_RNE", scale=orc.scale_from_amax(orc.amax(x), 448.0))
    for k in ("b", "c"):
        got = np.concatenate([p[k].numpy() for p in parts])
        assert_bits_equal(got, q(w[k]), "weight %s: the scale of the WHOLE logical weight" % k)
    # the tied storage went through the batch once and then once more (the reference's loop quantises it per module):
    # the second pass is a per-tensor quantisation of the once-quantised logical weight, one collective of its own
    got = np.concatenate([p["a"].numpy() for p in parts])
    assert_bits_equal(got, q(q(w["a"])), "tied weight: q(q(w)), as the sequential loop")
    assert all(int(p["collectives"]) == 2 for p in parts), "one collective for the batch + one for the repeated storage"


def test_offset_guard_compares_on_the_device_word():
    from mpemu.sharded import _OffsetGuard

    g = _OffsetGuard("cpu")
    g.note(torch.tensor([5, -5], dtype=torch.int32))
    g.poll()
    g.check()
    g.note(torch.tensor([7, -5], dtype=torch.int32))   # max(v) = 7, max(-v) = -5: ranks held 5 and 7
    with pytest.raises(RuntimeError, match="disagree on the stochastic call offset"):
        g.poll()
    g.check()                                          # raised once, then clean again
    g.note(torch.tensor([0, 0], dtype=torch.int32))
    g.check()


def test_row_segments_cover_any_cut():
    from mpemu.sharded import row_segments

    for cols in (1, 3, 8, 40):
        for start in range(0, 50):
            for numel in range(0, 130):
                segs = row_segments(numel, cols, start)
                pos = start
                for lo, nr, nc, r0, c0 in segs:
                    assert lo == pos - start and r0 * cols + c0 == pos and nr >= 1 and 1 <= nc <= cols
                    assert nr == 1 or (c0 == 0 and nc == cols)
                    assert c0 + nc <= cols
                    pos += nr * nc
                assert pos == start + numel and len(segs) <= 3
