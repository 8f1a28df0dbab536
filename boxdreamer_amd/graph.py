"""HIP-graph capture of the whole hot path (encoder -> decoder -> corner decode) for a fixed shape.

The path is ~300 short-to-medium kernel launches per step; at small batch (the demo loop runs batch 1,
/root/reference/src/demo/demo.py:1451-1506) host launch cost and inter-kernel gaps dominate.  All entry points enqueue on the
caller's stream, never synchronise and never allocate, so the sequence is capturable as is: inputs are copied into static
buffers, the graph is replayed, outputs are read from static buffers.
"""
from __future__ import annotations

import torch

from . import hip_ops


class GraphedPath:
    def __init__(self, encoder, decoder, B: int, T: int, size: int = 224, in_dtype: torch.dtype = torch.bfloat16,
                 device=None, want_idx: bool = False, capture_error_mode: str = "global"):
        self.encoder, self.decoder = encoder, decoder
        dev = torch.device("cuda") if device is None else torch.device(device)
        self.images = torch.zeros((B, T, 3, size, size), dtype=in_dtype, device=dev)
        self.bbox_feat = torch.zeros((B, T, 8, size, size), dtype=in_dtype, device=dev)
        self.mask = torch.zeros((B, T), dtype=torch.bool, device=dev)
        self.mask[:, T - 1] = True
        self.want_idx = want_idx
        # sub-batch lanes (the modules' `hip_lanes`): the library's side streams / events exist before the capture starts
        from . import _lib
        with torch.cuda.device(dev.index if dev.index is not None else torch.cuda.current_device()):
            _lib.check(_lib.load().bd_lanes_prepare(), "bd_lanes_prepare")
        # warm-up on a side stream (allocates workspaces / packs weights outside the capture), then capture
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(2):
                self._run()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        # capture_error_mode: torch's default ("global") makes unsafe CUDA calls of OTHER host threads -- an allocation, say -- invalidate
        # this capture; a process whose other threads keep working on the device passes "thread_local" (the library's own side streams
        # are per host thread: include/boxdreamer_hip.h, Sub-batch lanes)
        with torch.cuda.graph(self.graph, capture_error_mode=capture_error_mode):
            self.out = self._run()
        # The captured launches hold RAW device pointers into memory the modules own (workspaces allocated during the
        # warm-up above, packed weights).  (1) Keep those tensors alive on this object, so nothing the modules do later can
        # turn a replay into a use-after-free; (2) mark the modules frozen (weak reference), so an operation that would
        # re-allocate or re-pack them -- a larger eager batch, .to(), a checkpoint load -- raises instead of leaving the
        # graph replaying on stale weights.  Deleting the GraphedPath lifts the freeze.
        enc_model = getattr(encoder, "model", encoder)
        self._pinned = [enc_model._ws, decoder._ws, list(enc_model._packed.values()), list(decoder._packed.values())]
        # every live graph on these modules freezes them (a second GraphedPath must not lift the first one's freeze)
        enc_model._frozen_by.add(self)
        decoder._frozen_by.add(self)

    def _run(self):
        # (BETR.forward skips its one-hot mask check -- a device sync -- by itself while the stream is capturing; the
        # module's `validate_inputs` flag is left alone, so eager use of the same decoder keeps the check)
        feats = self.encoder.predict(self.images)
        heat = self.decoder(self.bbox_feat, self.images, self.mask, feats, None)
        kp, kn, idx = hip_ops.decode_topk(heat, want_idx=self.want_idx)
        return heat, kp, kn, idx

    def set_inputs(self, images: torch.Tensor, bbox_feat: torch.Tensor, query_idx: torch.Tensor | None = None):
        self.images.copy_(images, non_blocking=True)
        self.bbox_feat.copy_(bbox_feat, non_blocking=True)
        if query_idx is not None:      # (a device comparison: an indexed assignment of the scalar True would upload it with a synchronising copy)
            T = self.mask.shape[1]
            self.mask.copy_(torch.arange(T, device=self.mask.device)[None, :] == query_idx.to(self.mask.device).long()[:, None])

    def replay(self):
        """Replays the captured step on the current stream; returns (heat, kp_px, kp_norm, idx) static tensors."""
        self.graph.replay()
        return self.out

    def __call__(self, images, bbox_feat, query_idx=None):
        self.set_inputs(images, bbox_feat, query_idx)
        return self.replay()
