"""The group-aware entry points on the GPU, all through the C ABI:
fpemu_amax_rows + fpemu_cast_rows_amax (per-channel scaling whose rows are spread over ranks), fpemu_amax_multi +
fpemu_cast_multi_scaled (one collective for a list of weights), the device-side offset guard, and the two-process /
one-GPU runs of mpemu.sharded and mpemu.qutils over them (gloo stages the words through the host; with two GPUs the
RCCL test of tests/test_gpu_sharded.py covers the nccl branch)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import orc
from _util import assert_bits_equal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip():
    from mpemu.pytquant.hip import _binding

    _binding.lib()
    return _binding


def _matrix(rows, cols, seed=41):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((rows, cols)) * np.exp(rng.standard_normal((rows, 1)) * 2.0)).astype(np.float32)
    if rows > 1:
        x[1] = 0.0
    x[rows - 1, cols - 1] = 777.0
    x[rows // 2, cols - 1] = -555.0
    return x


def _rows_oracle(x, mode, flt_max, clamp=True, seed=0, offset=0):
    out = np.empty_like(x)
    for r in range(x.shape[0]):
        s = orc.scale_from_amax(orc.amax(x[r]), flt_max, clamp=clamp)
        out[r] = orc.cast(x[r], mode, scale=s, seed=seed, offset=offset + r)
    return out


SHAPES = [(300, 2), (64, 16), (257, 100), (33, 1024), (9, 4096), (5, 8200), (3, 20000), (2, 70000), (7, 333), (1000, 7)]


@pytest.mark.parametrize("rows,cols", SHAPES)
@pytest.mark.parametrize("mode,flt_max", [("E4M3_RNE", 448.0), ("E5M2_STOCHASTIC", 57344.0)])
def test_amax_rows_then_cast_rows_amax_equals_quantize_rows(hip, rows, cols, mode, flt_max):
    fmt, rm = hip.parse_mode(mode)
    x = _matrix(rows, cols)
    t = torch.from_numpy(x).to(DEV)
    want = torch.empty_like(t)
    hip.quantize_rows_(t, want, rows, fmt, rm, flt_max, offset=17)
    words = torch.full((rows,), -1, dtype=torch.int32, device=DEV)
    hip.amax_rows_(t, rows, words)
    assert np.array_equal(words.cpu().numpy().view(np.float32), np.abs(x).max(axis=1))
    got = torch.empty_like(t)
    hip.cast_rows_amax_(t, got, rows, fmt, rm, words, flt_max, seed=None, offset=17)
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    if rows * cols <= 300000:
        seed = hip.get_rng_state()[0]
        assert_bits_equal(got.cpu().numpy(), _rows_oracle(x, mode, flt_max, seed=seed, offset=17), "%s %dx%d" % (mode, rows, cols))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_amax_rows_16bit_storage(hip, dtype):
    fmt, rm = hip.parse_mode("E4M3_RNE")
    for rows, cols in ((40, 24), (9, 4096), (3, 9000)):
        t = torch.from_numpy(_matrix(rows, cols)).to(DEV).to(dtype)
        want = torch.empty_like(t)
        hip.quantize_rows_(t, want, rows, fmt, rm, 448.0)
        words = torch.zeros(rows, dtype=torch.int32, device=DEV)
        hip.amax_rows_(t, rows, words)
        assert torch.equal(words.view(torch.float32), t.float().abs().max(dim=1).values)
        got = torch.empty_like(t)
        hip.cast_rows_amax_(t, got, rows, fmt, rm, words, 448.0)
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize("mode,flt_max", [("E4M3_RNE", 448.0), ("E5M2_STOCHASTIC", 57344.0), ("E3M4_STOCHASTIC", 30.0)])
@pytest.mark.parametrize("rows,cols,cut", [(12, 4096, 1024), (12, 4096, 1500), (6, 9000, 4097), (50, 96, 40), (4, 70000, 2048)])
def test_row_segments_with_a_column_base_equal_the_whole_rows(hip, rows, cols, cut, mode, flt_max):
    """Columns [cut, cols) of every row as their own [rows, cols - cut] buffer, cast with the whole rows' amax words
    and col_base = cut, are bit for bit those columns of the whole tensor's per-channel quantisation."""
    fmt, rm = hip.parse_mode(mode)
    x = _matrix(rows, cols)
    t = torch.from_numpy(x).to(DEV)
    whole = torch.empty_like(t)
    hip.quantize_rows_(t, whole, rows, fmt, rm, flt_max, offset=5)
    words = torch.zeros(rows, dtype=torch.int32, device=DEV)
    hip.amax_rows_(t, rows, words)
    for c0, c1 in ((0, cut), (cut, cols)):
        part = t[:, c0:c1].contiguous()
        got = torch.empty_like(part)
        hip.cast_rows_amax_(part, got, rows, fmt, rm, words, flt_max, offset=5, col_base=c0)
        assert torch.equal(got.view(torch.int32), whole[:, c0:c1].contiguous().view(torch.int32)), (c0, c1)


def test_amax_multi_then_cast_multi_scaled_equals_quantize_per_tensor_multi(hip):
    rng = np.random.default_rng(3)
    sizes = [1, 5, 777, 4096, 100000, 0, 33333, 1 << 20] + [1000 + 37 * i for i in range(40)]
    for mode, flt_max in (("E4M3_RNE", 448.0), ("E5M2_STOCHASTIC", 57344.0)):
        fmt, rm = hip.parse_mode(mode)
        xs = [torch.from_numpy((rng.standard_normal(n) * (i + 1)).astype(np.float32)).to(DEV) for i, n in enumerate(sizes)]
        hip.set_rng_state((99, 7))
        want, ws = hip.quantize_per_tensor_multi_(xs, fmt, rm, flt_max, dsts=[torch.empty_like(t) for t in xs])
        words = torch.full((len(xs),), -1, dtype=torch.int32, device=DEV)
        hip.amax_multi_(xs, words)
        w_host = words.cpu().numpy().view(np.float32)
        for i, t in enumerate(xs):
            assert w_host[i] == (float(t.abs().max()) if t.numel() else 0.0), i
        hip.set_rng_state((99, 7))
        got = hip.cast_multi_scaled_(xs, fmt, rm, words, flt_max, dsts=[torch.empty_like(t) for t in xs])
        assert hip.get_rng_state() == (99, 7 + (len(xs) if mode.endswith("STOCHASTIC") else 0))
        for i, (g, w) in enumerate(zip(got, want)):
            assert torch.equal(g.view(torch.int32), w.view(torch.int32)), (mode, i, sizes[i])


def test_offset_guard_on_the_device():
    from mpemu.sharded import _OffsetGuard

    g = _OffsetGuard(DEV)
    g.note(torch.tensor([9, -9], dtype=torch.int32, device=DEV))
    g.poll()
    torch.cuda.synchronize()
    g.poll()          # the first copy has landed: clean
    g.check()
    g.note(torch.tensor([9, -8], dtype=torch.int32, device=DEV))
    g.poll()          # looks at the earlier (clean) copy, starts the next one
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="disagree on the stochastic call offset"):
        g.poll()      # non-blocking: the copy that saw the disagreement has landed
    g.check()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Tiny(torch.nn.Module):
    def __init__(self, shard_rows):
        super().__init__()
        self.a = torch.nn.Linear(240, shard_rows)
        self.b = torch.nn.Linear(1600, shard_rows)
        self.c = torch.nn.Conv2d(64, shard_rows, 3)
        self.tied = torch.nn.Linear(240, shard_rows)
        self.tied.weight = self.a.weight


def _logical_weights(total_rows):
    rng = np.random.default_rng(5)
    w = {"a": rng.standard_normal((total_rows, 240)), "b": rng.standard_normal((total_rows, 1600)) * 3.0,
         "c": rng.standard_normal((total_rows, 64, 3, 3)) * 0.1}
    w = {k: v.astype(np.float32) for k, v in w.items()}
    w["a"][total_rows - 1, 3] = 9.5
    w["b"][total_rows - 1, 0] = -40.0
    w["c"][total_rows - 1, 0, 0, 0] = 2.25
    return w


def _cut(n, world, rank, align=2048):
    per = -(-n // world)
    per = -(-per // align) * align
    a = min(n, rank * per)
    return a, min(n, a + per)


def _worker(rank, world, port, rows, cols, outdir):
    for p in (os.path.join(ROOT, "fp8-emulation-toolkit_amd"), os.path.join(ROOT, "oracle"),
              os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from mpemu import qutils
    from mpemu.pytquant.hip import _binding as hip
    from mpemu.sharded import ShardedCaster

    x = _matrix(rows, cols).reshape(-1)
    a, b = _cut(x.size, world, rank)
    shard = torch.from_numpy(x[a:b].copy()).to(DEV)
    sc = ShardedCaster()
    res = {}
    hip.set_rng_state((77, 30))
    out, words = sc.quantize_rows_flat(shard, rows, cols, a, "E5M2_STOCHASTIC", 57344.0)
    res["sr"], res["words"] = out.cpu(), words.cpu()
    qutils.set_amax_process_group(dist.group.WORLD, elem_base=a, layout="flat", logical=(rows, cols))
    res["qt"] = qutils.quantize_tensor(shard, qutils.TensorQuantConfig("e4m3", "rne", "per-channel"), None).cpu()
    res["qt_sr"] = qutils.quantize_tensor(shard, qutils.TensorQuantConfig("e4m3", "stochastic", "per-channel"), None).cpu()
    res["state"] = torch.tensor(hip.get_rng_state()[1])
    qutils.set_amax_process_group(None)

    # quantize_model_weights under a group: every weight a dim-0 shard of a logical weight
    srows = 6
    w = _logical_weights(world * srows)
    m = _Tiny(srows).to(DEV)
    with torch.no_grad():
        for k in ("a", "b", "c"):
            getattr(m, k).weight.copy_(torch.from_numpy(w[k][rank * srows:(rank + 1) * srows]))
    cfg = qutils.ModuleQuantConfig(wt_qconfig=qutils.TensorQuantConfig("e4m3", "rne", "per-tensor"))
    qutils.set_amax_process_group(dist.group.WORLD)
    qutils.quantize_model_weights(m, {n: cfg for n in ("a", "b", "c", "tied")})
    qutils.set_amax_process_group(None)
    for k in ("a", "b", "c"):
        res["w_" + k] = getattr(m, k).weight.detach().cpu()
    torch.cuda.synchronize()
    torch.save(res, os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("rows,cols", [(37, 1000), (5, 30000), (3, 150000)])
def test_two_processes_one_gpu_flat_rows_and_model_weights(tmp_path, rows, cols):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), rows, cols, str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=True) for r in range(world)]
    x = _matrix(rows, cols)
    cat = lambda k: np.concatenate([p[k].numpy().reshape(-1) for p in parts]).reshape(rows, cols)
    assert_bits_equal(cat("sr"), _rows_oracle(x, "E5M2_STOCHASTIC", 57344.0, seed=77, offset=30), "flat rows, SR")
    assert_bits_equal(cat("qt"), _rows_oracle(x, "E4M3_RNE", 448.0), "quantize_tensor per-channel, layout flat")
    assert_bits_equal(cat("qt_sr"), _rows_oracle(x, "E4M3_STOCHASTIC", 448.0, seed=77, offset=30 + rows),
                      "quantize_tensor per-channel stochastic, layout flat")
    assert all(int(p["state"]) == 30 + 2 * rows for p in parts)
    for p in parts:
        assert np.array_equal(p["words"].numpy().view(np.float32), np.abs(x).max(axis=1))
    w = _logical_weights(world * 6)
    q = lambda v: orc.cast(v, "E4M3_RNE", scale=orc.scale_from_amax(orc.amax(v), 448.0))
    for k in ("b", "c"):
        assert_bits_equal(np.concatenate([p["w_" + k].numpy() for p in parts]), q(w[k]), "weight " + k)
    assert_bits_equal(np.concatenate([p["w_a"].numpy() for p in parts]), q(q(w["a"])), "tied weight: q(q(w))")


def test_tied_weights_single_device_equal_the_sequential_loop():
    """quantize_model_weights de-duplicates storage: a weight shared by two modules ends as q(q(w)),
    what the reference's sequential loop produces -- not as a race between two sets of workgroups."""
    from mpemu import qutils

    torch.manual_seed(1)
    m = _Tiny(48).to(DEV)
    w0 = {k: getattr(m, k).weight.detach().cpu().numpy().copy() for k in ("a", "b", "c")}
    for scheme in ("rne", "rtz"):
        with torch.no_grad():
            for k in w0:
                getattr(m, k).weight.copy_(torch.from_numpy(w0[k]))
        cfg = qutils.ModuleQuantConfig(wt_qconfig=qutils.TensorQuantConfig("e5m2", scheme, "per-tensor"))
        qutils.quantize_model_weights(m, {n: cfg for n in ("a", "b", "c", "tied")})
        mode = "E5M2_" + scheme.upper()
        q = lambda v: orc.cast(v, mode, scale=orc.scale_from_amax(orc.amax(v), 57344.0))
        assert_bits_equal(m.a.weight.detach().cpu().numpy(), q(q(w0["a"])), "tied, " + scheme)
        assert m.tied.weight.data_ptr() == m.a.weight.data_ptr()
        for k in ("b", "c"):
            assert_bits_equal(getattr(m, k).weight.detach().cpu().numpy(), q(w0[k]), k)
