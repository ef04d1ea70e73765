"""Sharded cast on the GPU: the HIP path must give the same bits whether a tensor is cast whole
or as shards (elem_base carries the global index for the stochastic noise; per-tensor scaling
uses the max-reduced amax word).  The 2-process case shares the one GPU of the test box and
uses gloo for the 4-byte all-reduce (RCCL refuses two ranks on one device; mpemu.sharded stages the
words through the host for non-nccl backends); the data path is the real HIP kernels.  With two or more
GPUs visible the same flow runs over the "nccl" backend (RCCL), one rank per GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import orc
from _util import assert_bits_equal, mixed_inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _data(n):
    x = mixed_inputs(n, seed=33)
    x[~np.isfinite(x) | (np.abs(x) > 1000.0)] = 0.25
    x[n - 7] = -2500.0
    return x


@pytest.mark.parametrize("mode,flt_max", [("E4M3_RNE", 448.0), ("E5M2_STOCHASTIC", 57344.0),
                                          ("E3M4_STOCHASTIC", 30.0), ("BFLOAT16_STOCHASTIC", 1.0)])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_single_process_shards_equal_whole(mode, flt_max, world):
    from mpemu.sharded import ShardedCaster, shard_bounds

    n = (1 << 20) + 4321
    x = _data(n)
    t = torch.from_numpy(x).cuda()
    sc = ShardedCaster()
    whole_unscaled = sc.cast(t, mode, seed=3, offset=5)
    word = sc.amax_words([t])
    whole_scaled = sc.quantize_per_tensor(t, mode, flt_max, seed=3, offset=6, amax_word=word)
    outs_u, outs_s, words = [], [], []
    for r in range(world):
        a, b = shard_bounds(n, world, r)
        words.append(sc.amax_words([t[a:b]]))
    gword = torch.stack(words).max(dim=0).values  # what the all-reduce(MAX) computes
    assert torch.equal(gword, word)
    for r in range(world):
        a, b = shard_bounds(n, world, r)
        outs_u.append(sc.cast(t[a:b], mode, elem_base=a, seed=3, offset=5))
        outs_s.append(sc.quantize_per_tensor(t[a:b], mode, flt_max, elem_base=a, seed=3, offset=6, amax_word=gword))
    assert torch.equal(torch.cat(outs_u).view(torch.int32), whole_unscaled.view(torch.int32))
    assert torch.equal(torch.cat(outs_s).view(torch.int32), whole_scaled.view(torch.int32))
    assert_bits_equal(whole_unscaled.cpu().numpy(), orc.cast(x, mode, seed=3, offset=5), mode)
    if not mode.startswith("BFLOAT16"):
        s = orc.scale_from_amax(orc.amax(x), flt_max)
        assert_bits_equal(whole_scaled.cpu().numpy(), orc.cast(x, mode, scale=s, seed=3, offset=6), mode + " scaled")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, outdir):
    for p in (os.path.join(ROOT, "fp8-emulation-toolkit_amd"), os.path.join(ROOT, "oracle"),
              os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mpemu.sharded import ShardedCaster, shard_bounds

    x = _data(n)
    a, b = shard_bounds(n, world, rank)
    shard = torch.from_numpy(x[a:b].copy()).cuda()
    sc = ShardedCaster()
    out = sc.quantize_per_tensor(shard, "E4M3_RNE", 448.0, elem_base=a)
    out_sr = sc.quantize_per_tensor(shard, "E5M2_STOCHASTIC", 57344.0, elem_base=a, seed=8, offset=2)
    torch.cuda.synchronize()
    torch.save({"q": out.cpu(), "sr": out_sr.cpu()}, os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_one_gpu_global_amax(tmp_path):
    n = 300000
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=True) for r in range(world)]
    x = _data(n)
    s = orc.scale_from_amax(orc.amax(x), 448.0)
    assert_bits_equal(np.concatenate([p["q"].numpy() for p in parts]), orc.cast(x, "E4M3_RNE", scale=s))
    s5 = orc.scale_from_amax(orc.amax(x), 57344.0)
    assert_bits_equal(np.concatenate([p["sr"].numpy() for p in parts]),
                      orc.cast(x, "E5M2_STOCHASTIC", scale=s5, seed=8, offset=2))


def _qutils_worker(rank, world, port, n, outdir, backend):
    for p in (os.path.join(ROOT, "fp8-emulation-toolkit_amd"), os.path.join(ROOT, "oracle"),
              os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from mpemu import qutils
    from mpemu.pytquant.hip import _binding as hip
    from mpemu.sharded import ShardedCaster, shard_bounds

    x = _data(n)
    a, b = shard_bounds(n, world, rank)
    shard = torch.from_numpy(x[a:b].copy()).to(dev)
    # (1) through the hooks' entry point: quantize_tensor under set_amax_process_group, stochastic mode
    qutils.set_amax_process_group(dist.group.WORLD, elem_base=lambda t: a)
    hip.set_rng_state((1234, 20))
    cfg = qutils.TensorQuantConfig("e5m2", "stochastic", "per-tensor")
    q_sr = qutils.quantize_tensor(shard, cfg, None)                     # offset 20
    q_sr2 = qutils.quantize_tensor(shard, cfg, None)                    # offset 21, back to back
    q_rne = qutils.quantize_tensor(shard, qutils.TensorQuantConfig("e4m3", "rne", "per-tensor"), None)
    qutils.set_amax_process_group(None)
    # (2) ShardedCaster: batched words, many collectives back to back
    sc = ShardedCaster()
    for _ in range(20):
        words = sc.amax_words([shard, shard * 0.5])
    out = sc.quantize_per_tensor(shard, "E4M3_RNE", 448.0, elem_base=a)
    torch.cuda.synchronize()
    torch.save({"sr": q_sr.cpu(), "sr2": q_sr2.cpu(), "rne": q_rne.cpu(), "q": out.cpu(), "words": words.cpu()},
               os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _check_qutils_parts(parts, n):
    x = _data(n)
    cat = lambda k: np.concatenate([p[k].numpy() for p in parts])
    s5 = orc.scale_from_amax(orc.amax(x), 57344.0)
    s4 = orc.scale_from_amax(orc.amax(x), 448.0)
    assert_bits_equal(cat("sr"), orc.cast(x, "E5M2_STOCHASTIC", scale=s5, seed=1234, offset=20), "quantize_tensor SR")
    assert_bits_equal(cat("sr2"), orc.cast(x, "E5M2_STOCHASTIC", scale=s5, seed=1234, offset=21), "second SR call")
    assert_bits_equal(cat("rne"), orc.cast(x, "E4M3_RNE", scale=s4), "quantize_tensor RNE")
    assert_bits_equal(cat("q"), orc.cast(x, "E4M3_RNE", scale=s4), "ShardedCaster")
    for p in parts:
        assert p["words"].view(torch.float32).tolist() == [2500.0, 1250.0]


def test_two_processes_one_gpu_quantize_tensor_group_path(tmp_path):
    """The hooks' entry point under set_amax_process_group, stochastic mode, two ranks sharing the GPU over gloo
    (host-staged words): the bits of the single-process cast of the whole tensor."""
    n, world = 300000, 2
    mp.spawn(_qutils_worker, args=(world, _free_port(), n, str(tmp_path), "gloo"), nprocs=world, join=True)
    parts = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=True) for r in range(world)]
    _check_qutils_parts(parts, n)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: fewer than 2 GPUs visible")
def test_rccl_one_rank_per_gpu(tmp_path):
    """The first run of the real thing wherever >= 2 GPUs are visible: backend "nccl" (= RCCL), the amax words
    all-reduced on the device in stream order."""
    n = 300000
    world = min(torch.cuda.device_count(), 8)
    mp.spawn(_qutils_worker, args=(world, _free_port(), n, str(tmp_path), "nccl"), nprocs=world, join=True)
    parts = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=True) for r in range(world)]
    _check_qutils_parts(parts, n)


def _one_rank_rccl_worker(rank, world, port, n, outdir):
    """A one-rank RCCL communicator on the box's GPU with the single-rank shortcut off: every collective of the
    product's nccl branch really goes through torch -> RCCL (all-reduce(MAX) of int32 words on the device, in stream
    order), and the device-side offset guard runs with its pinned word and event polling."""
    for p in (os.path.join(ROOT, "fp8-emulation-toolkit_amd"), os.path.join(ROOT, "oracle"),
              os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from mpemu import qutils, sharded
    from mpemu.pytquant.hip import _binding as hip

    sharded._SINGLE_RANK_SHORTCUT = False
    assert sharded.group_backend() == "nccl"
    x = _data(n)
    t = torch.from_numpy(x).to(dev)
    res = {}
    # plain words, back to back
    w = torch.tensor([5, 0x45200000, 0], dtype=torch.int32, device=dev)
    for _ in range(30):
        sharded.all_reduce_max_words(w)
    res["plain"] = w.cpu()
    # the hooks' entry point under the group: per-tensor stochastic (offset guard), deterministic, per-channel rows
    qutils.set_amax_process_group(dist.group.WORLD, elem_base=lambda tt: 0)
    hip.set_rng_state((1234, 20))
    cfg = qutils.TensorQuantConfig("e5m2", "stochastic", "per-tensor")
    res["sr"] = qutils.quantize_tensor(t, cfg, None).cpu()                       # offset 20
    for _ in range(12):                                                          # the guard's polls land
        qutils.quantize_tensor(t, cfg, None)
    res["rne"] = qutils.quantize_tensor(t, qutils.TensorQuantConfig("e4m3", "rne", "per-tensor"), None).cpu()
    qutils.set_amax_process_group(None)
    sharded.check_offsets()
    sc = sharded.ShardedCaster()
    res["q"] = sc.quantize_per_tensor(t, "E4M3_RNE", 448.0).cpu()
    rows, cols = 100, n // 100
    flat = t[:rows * cols].contiguous()
    out, words = sc.quantize_rows_flat(flat, rows, cols, 0, "E4M3_STOCHASTIC", 448.0, seed=5, offset=40)
    res["rows"] = out.cpu()
    outs, _w = sc.quantize_per_tensor_multi([t.clone(), (t * 0.5)], "E4M3_RNE", 448.0)
    res["multi0"], res["multi1"] = outs[0].cpu(), outs[1].cpu()
    sharded.check_offsets()
    torch.cuda.synchronize()
    torch.save(res, os.path.join(outdir, "one.pt"))
    dist.destroy_process_group()


def test_rccl_one_rank_communicator_drives_the_nccl_branch(tmp_path):
    n = 300000
    mp.spawn(_one_rank_rccl_worker, args=(1, _free_port(), n, str(tmp_path)), nprocs=1, join=True)
    res = torch.load(os.path.join(str(tmp_path), "one.pt"), weights_only=True)
    x = _data(n)
    assert res["plain"].tolist() == [5, 0x45200000, 0]
    s5 = orc.scale_from_amax(orc.amax(x), 57344.0)
    s4 = orc.scale_from_amax(orc.amax(x), 448.0)
    assert_bits_equal(res["sr"].numpy(), orc.cast(x, "E5M2_STOCHASTIC", scale=s5, seed=1234, offset=20), "SR")
    assert_bits_equal(res["rne"].numpy(), orc.cast(x, "E4M3_RNE", scale=s4), "RNE")
    assert_bits_equal(res["q"].numpy(), orc.cast(x, "E4M3_RNE", scale=s4), "ShardedCaster")
    assert_bits_equal(res["multi0"].numpy(), orc.cast(x, "E4M3_RNE", scale=s4), "multi 0")
    h = (x * np.float32(0.5)).astype(np.float32)
    assert_bits_equal(res["multi1"].numpy(), orc.cast(h, "E4M3_RNE", scale=orc.scale_from_amax(orc.amax(h), 448.0)),
                      "multi 1")
    rows, cols = 100, n // 100
    got = res["rows"].numpy().reshape(rows, cols)
    for r in (0, 1, 57, 99):
        row = x[r * cols:(r + 1) * cols]
        s = orc.scale_from_amax(orc.amax(row), 448.0)
        assert_bits_equal(got[r], orc.cast(row, "E4M3_STOCHASTIC", scale=s, seed=5, offset=40 + r), "row %d" % r)
