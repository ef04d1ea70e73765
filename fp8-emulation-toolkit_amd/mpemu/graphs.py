"""mpemu.graphs -- replay a hooked forward pass as ONE HIP graph launch.

The emulation hooks turn every layer into a handful of small launches (amax, cast, ...): at inference
batch sizes the forward pass is launch-bound (BERT-base 8x128: 3.0 ms plain, 4.9 ms hooked, most of the
difference host time).  Every native entry point of this package only enqueues work on the current
stream -- no allocation, no synchronisation, no host read-back -- so a whole hooked forward captures into
a HIP graph and replays with a single launch.  No counterpart in the reference (its casts synchronise
the device on every call).

    model, emu = mpt_emu.quantize_model(model, dtype="e4m3", device="cuda")
    fwd = GraphedForward(model, example_ids)        # warm-up + capture
    out = fwd(ids)                                   # copy in, replay, static output tensors

Limits: inference only (no autograd through the replay); shapes are those of the example inputs;
stochastic-rounding modes replay the noise of the captured call (seed / offset are kernel arguments), so
use it with deterministic modes; outputs are static buffers overwritten by the next call.  The casts
replay bit-identically; torch's own kernels may not (HF BERT's attention takes another kernel path under
capture, 1.7e-6 apart already without hooks), so compare graph against eager per module, not per model.
"""
import torch

from .pytquant.hip import _binding as _hip


def _flatten(out):
    if isinstance(out, torch.Tensor):
        return [out]
    if isinstance(out, (list, tuple)):
        return [t for o in out for t in _flatten(o)]
    if isinstance(out, dict):
        return [t for o in out.values() for t in _flatten(o)]
    if hasattr(out, "to_tuple"):          # transformers ModelOutput
        return _flatten(out.to_tuple())
    return []


class GraphedForward(object):
    def __init__(self, model, *example_inputs, warmup=3, **example_kwargs):
        if not all(t.is_cuda for t in example_inputs if isinstance(t, torch.Tensor)):
            raise RuntimeError("GraphedForward: example inputs must live on the GPU")
        self.model = model
        self._static_in = [t.clone() if isinstance(t, torch.Tensor) else t for t in example_inputs]
        self._kwargs = example_kwargs
        dev = next((t.device for t in example_inputs if isinstance(t, torch.Tensor)), torch.device("cuda"))
        # this graph's own scratch for the per-tensor quantisations, allocated BEFORE capture: the graph bakes
        # its address in, so it must neither come from another graph's pool nor be shared with eager calls
        self._ws = torch.empty(_hip._WS_WORDS, dtype=torch.int32, device=dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad(), _hip.use_workspace(self._ws):
            for _ in range(warmup):       # lazy initialisation (hipBLASLt handles, ...) outside capture
                model(*self._static_in, **self._kwargs)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph), _hip.use_workspace(self._ws):
            self._static_out = model(*self._static_in, **self._kwargs)
        self._flat_out = _flatten(self._static_out)

    def __call__(self, *inputs):
        if len(inputs) != len(self._static_in):
            raise ValueError("GraphedForward: expected %d inputs" % len(self._static_in))
        for dst, src in zip(self._static_in, inputs):
            if isinstance(dst, torch.Tensor):
                if src.shape != dst.shape or src.dtype != dst.dtype:
                    raise ValueError("GraphedForward: input shape / dtype differs from the captured example")
                dst.copy_(src)
        self.graph.replay()
        return self._static_out
