"""Model definitions shared by tests/golden/make_golden.py (run against the REFERENCE's mpemu package) and the
tests (run against this repo's): a module holding every whitelisted type, a tiny MLP, and a 2-layer tiny BERT
whose attention products go through the module_wrappers (Matmul / BatchMatmul / EltwiseAdd) so the hooks
attach to them.  `from mpemu.module_wrappers import ...` resolves to whichever mpemu is on sys.path.
(EltwiseMul is left out: the reference's constructor raises NameError, module_wrappers/eltwise.py:34.)"""
import math

import torch
from torch import nn

from mpemu.module_wrappers import AddMatmul, BatchMatmul, EltwiseAdd, EltwiseDiv, Matmul


class Zoo(nn.Module):
    """One instance of each whitelisted module type; never run, only configured."""

    def __init__(self):
        super().__init__()
        self.conv_small = nn.Conv2d(3, 8, 3)          # in / out channels < 64: never op-patched
        self.conv_big = nn.Conv2d(64, 64, 3)
        self.lin1 = nn.Linear(16, 16)
        self.lin2 = nn.Linear(16, 4)
        self.emb = nn.Embedding(10, 16)
        self.embbag = nn.EmbeddingBag(10, 16)
        self.ln = nn.LayerNorm(16)
        self.mm = Matmul()
        self.bmm = BatchMatmul()
        self.addmm = AddMatmul()
        self.add = EltwiseAdd()
        self.div = EltwiseDiv()
        self.relu = nn.ReLU()                          # not whitelisted
        self.block = nn.Sequential(nn.Linear(4, 4), nn.LayerNorm(4))


def tiny_mlp():
    torch.manual_seed(4321)
    return nn.Sequential(nn.Linear(24, 48), nn.ReLU(), nn.Linear(48, 10))


class TinyBertLayer(nn.Module):
    def __init__(self, d, heads, ff):
        super().__init__()
        self.h, self.dh = heads, d // heads
        self.q, self.k, self.v, self.o = nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d)
        self.scores = BatchMatmul()      # Q K^T
        self.context = BatchMatmul()     # P V
        self.res1, self.res2 = EltwiseAdd(), EltwiseAdd()
        self.ln1, self.ln2 = nn.LayerNorm(d), nn.LayerNorm(d)
        self.ff1, self.ff2 = nn.Linear(d, ff), nn.Linear(ff, d)

    def forward(self, x):
        B, S, D = x.shape
        split = lambda t: t.view(B, S, self.h, self.dh).transpose(1, 2).reshape(B * self.h, S, self.dh)
        q, k, v = split(self.q(x)), split(self.k(x)), split(self.v(x))
        p = torch.softmax(self.scores(q, k.transpose(1, 2).contiguous()) / math.sqrt(self.dh), dim=-1)
        c = self.context(p, v).view(B, self.h, S, self.dh).transpose(1, 2).reshape(B, S, D)
        x = self.ln1(self.res1(x, self.o(c)))
        return self.ln2(self.res2(x, self.ff2(torch.nn.functional.gelu(self.ff1(x)))))


class TinyBert(nn.Module):
    def __init__(self, vocab=50, d=32, heads=4, ff=64, layers=2, seq=12):
        super().__init__()
        self.embeddings = nn.Embedding(vocab, d)
        self.pos = nn.Parameter(torch.randn(seq, d) * 0.1)
        self.ln0 = nn.LayerNorm(d)
        self.layers = nn.ModuleList([TinyBertLayer(d, heads, ff) for _ in range(layers)])
        self.pool = Matmul()             # [B, S, D] x [D, 2]: a functional matmul behind a hookable module
        self.head_w = nn.Parameter(torch.randn(d, 2) * 0.2)

    def forward(self, ids):
        x = self.ln0(self.embeddings(ids) + self.pos)
        for l in self.layers:
            x = l(x)
        return self.pool(x, self.head_w)


def tiny_bert():
    torch.manual_seed(777)
    m = TinyBert()
    ids = torch.randint(0, 50, (3, 12), generator=torch.Generator().manual_seed(5))
    return m, ids
